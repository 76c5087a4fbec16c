/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see ora_util.h).
 *
 * wtgbo_oracle — `wtgbo` on the CPU: the sequential side (options, overlap graph, candidate walks, commit order) is the
 * product's own host code (smartdenovo_amd/csrc/host/wtgbo_core.h — integer bookkeeping with no device in it), the pair alignment is
 * the oracle's restatement of align_hzmaux (ora_hzmaux.h) instead of the device pipeline.  Pinned: its output equals the goldens of the
 * real `wtgbo -t 1` (tests/test_oracle_golden.py::test_wtgbo_oracle_equals_reference_golden).  Used by tests and by nothing else.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "ora_hzmaux.h"
#include "../smartdenovo_amd/csrc/host/wtgbo_core.h"

typedef struct { ora_hzmaux_t A; ora_auxparams_t P; uint32_t indexed; vec_u8 t, q; } gbo_cpu_t;

static void unpack_read(const gbo_t *G, uint32_t id, int rev, vec_u8 *dst){
	const uint64_t off = G->st.reads[id].off; const uint32_t len = G->st.reads[id].len;
	vec_u8_reserve(dst, (size_t)len + 8); dst->n = len;
	for(uint32_t i = 0; i < len; i++){
		const uint64_t x = rev ? off + len - 1 - i : off + i;
		const unsigned b = (unsigned)(G->st.bits[x >> 5] >> (((~x) & 31u) << 1)) & 3u;
		dst->a[i] = (uint8_t)(rev ? 3u - b : b);
	}
}

static void gbo_align_jobs(gbo_t *G, const gbo_job_t *jobs, size_t n, gbo_res_t *res){
	gbo_cpu_t *D = (gbo_cpu_t*)G->backend;
	for(size_t i = 0; i < n; i++){
		gbo_res_t *r = &res[i]; memset(r, 0, sizeof *r);
		if(D->indexed != jobs[i].obj){ unpack_read(G, jobs[i].obj, 0, &D->t); ora_hzmaux_index(&D->A, &D->P, D->t.a, (uint32_t)D->t.n); D->indexed = jobs[i].obj; }
		unpack_read(G, jobs[i].qry, (int)jobs[i].dir, &D->q);
		if(!ora_align_hzmaux(&D->A, &D->P, D->q.a, (int)D->q.n, G->O.refine, G->O.min_id)) continue;
		const ora_aln_t x = D->A.hit;
		r->ok = 1; r->score = x.score; r->tb = x.tb; r->te = x.te; r->qb = x.qb; r->qe = x.qe; r->aln = x.aln; r->mat = x.mat; r->mis = x.mis; r->ins = x.ins; r->del = x.del;
		uint32_t *dst = gbo_cigar_space(G, D->A.cigars.n);
		memcpy(dst, D->A.cigars.a, 4 * D->A.cigars.n);
		r->cig_off = (uint64_t)(dst - G->cigar_pool); r->cig_len = (uint32_t)D->A.cigars.n;
	}
}

int main(int argc, char **argv){
	gbo_t *G = (gbo_t*)calloc(1, sizeof(gbo_t));
	if(gbo_parse_args(&G->O, argc, argv)) return gbo_usage();
	gbo_opt_t *o = &G->O;
	G->st.keep_text = 0;        /* the CPU checker packs while reading (its lrand48 draws are the C library's own) */
	gbo_load_inputs(G);
	static gbo_cpu_t D;
	D.indexed = 0xFFFFFFFFu;
	D.P.zsize = (uint32_t)o->zsize; D.P.hz = (uint32_t)o->hz; D.P.zwin = (uint32_t)o->kwin; D.P.zstep = (uint32_t)o->kstep; D.P.zovl = (uint32_t)o->zovl;
	D.P.zmax = (uint32_t)o->zcut; D.P.zvar = (uint32_t)o->kvar; D.P.w = o->w; D.P.W = o->W; D.P.ew = o->ew; D.P.rw = o->w;
	D.P.M = o->M; D.P.X = o->X; D.P.I = o->O; D.P.D = o->O; D.P.E = o->E; D.P.T = o->T;      /* wtgbo.c:470-486 */
	G->backend = &D;
	return gbo_run(G);
}
