/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see ora_util.h).
 *
 * wtext_oracle — `wtext` on the CPU: the sequential side (options, retained regions, CIGAR clipping and re-scoring, record order) is the
 * product's own host code (smartdenovo_amd/csrc/host/wtext_core.h — text and integer bookkeeping with no device in it), the end extensions are the
 * oracle's restatement of kswx_extend_align_shift_core (ora_sw.h, pinned against the reference routine by the DP vectors).  Its output must equal the
 * goldens of the real `wtext -t 1` and `oracle/_ref/wtext_ref` run live (tests/test_wtext.py) - which is what pins the host code, shared with the product.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ora_sw.h"
#include "../smartdenovo_amd/csrc/host/wtext_core.h"

static void unpack(const wx_t *W, uint32_t id, int rev, vec_u8 *dst){
	const uint32_t len = W->pblen[id];
	vec_u8_reserve(dst, (size_t)len + 8); dst->n = len;
	for(uint32_t i = 0; i < len; i++) dst->a[i] = (uint8_t)wx_base(W, id, rev, i);
}
static void wx_extend_jobs(wx_t *W, const wx_job_t *jobs, size_t n, wx_jobres_t *res){
	const wx_opt_t *o = &W->O;
	vec_u8 q, t; memset(&q, 0, sizeof q); memset(&t, 0, sizeof t);
	ora_swmem_t mem; memset(&mem, 0, sizeof mem);
	vec_u32 cg; memset(&cg, 0, sizeof cg);
	for(size_t i = 0; i < n; i++){
		const wx_job_t *j = &jobs[i];
		unpack(W, j->q_read, (int)j->q_rev, &q); unpack(W, j->t_read, (int)j->t_rev, &t);
		cg.n = 0;
		const ora_aln_t x = ora_extend_shift(j->q_len, q.a + (j->q_len > 0 ? j->q_from : 0), j->t_len, t.a + (j->t_len > 0 ? j->t_from : 0), j->strand, j->init_score, o->W, o->M, o->X, o->O, o->O, o->E, o->T, &mem, &cg);
		wx_jobres_t *r = &res[i];
		r->x.score = x.score; r->x.tb = x.tb; r->x.te = x.te; r->x.qb = x.qb; r->x.qe = x.qe; r->x.aln = x.aln; r->x.mat = x.mat; r->x.mis = x.mis; r->x.ins = x.ins; r->x.del = x.del;
		uint32_t *dst = wx_cigar_space(W, cg.n);
		memcpy(dst, cg.a, 4 * cg.n);
		r->cig_off = (uint64_t)(dst - W->cigar_pool); r->cig_len = (uint32_t)cg.n;
	}
}
int main(int argc, char **argv){
	wx_t *W = (wx_t*)calloc(1, sizeof(wx_t));
	if(wx_parse_args(&W->O, argc, argv)) return wx_usage();
	wx_load_inputs(W);
	return wx_run(W);
}
