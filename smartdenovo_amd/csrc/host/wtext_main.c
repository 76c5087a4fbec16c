/*
 * wtext_main.c — drop-in `wtext`: overlaps clipped to the retained regions of their reads (`-b`, e.g. wtobt's output) and extended
 * to the region ends.  Same options, same 17-column records as the reference's wtext.c; the output is that of `wtext -t 1`.
 *
 * Host (wtext_core.h, plain C): inputs, the clipping and re-scoring of every overlap's CIGAR, the order of the records.
 * Device (libwtzmo_hip.so, wtz_extend_batch): every end extension - kswx_extend_align (kswx.h:469-481) is kswx_extend_align_shift_core, the K-sw3
 * routine of wtzmo's stitched alignments; the jobs of a block go through the same dispatch (register DP on one / four wavefronts per job).
 * There is no CPU implementation of the extension in this program: without a HIP device it exits with an error.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "wtzmo_hip.h"
#include "wtext_core.h"

#define DIE_WTZ(rc, what) do { if((rc) != WTZ_OK){ fprintf(stderr, " -- %s failed: %s --\n", what, wtz_last_error()); WX_DIE(); } } while(0)

typedef struct { wtz_ctx_t *ctx; wtz_dp_problem_t *pr; wtz_dp_result_t *rs; size_t cap; uint32_t *cig; uint64_t capcig; int W; } wx_dev_t;
typedef struct { const wtz_params_c *P; int device; uint64_t pool_bytes; wtz_ctx_t *ctx; int rc; char err[256]; pthread_t th; } wx_ctxjob_t;
static void *wx_ctxjob_main(void *arg){
	wx_ctxjob_t *j = (wx_ctxjob_t*)arg;
	j->rc = wtz_ctx_create(j->device, j->P, j->pool_bytes, &j->ctx);
	if(j->rc != WTZ_OK) snprintf(j->err, sizeof j->err, "%s", wtz_last_error());
	return NULL;
}

static void wx_extend_jobs(wx_t *W, const wx_job_t *jobs, size_t n, wx_jobres_t *res){
	wx_dev_t *D = (wx_dev_t*)W->backend;
	size_t at = 0, step = n;
	while(at < n){
		if(step > n - at) step = n - at;
		if(step > D->cap){ D->cap = step; D->pr = (wtz_dp_problem_t*)hx_realloc(D->pr, sizeof(wtz_dp_problem_t) * step); D->rs = (wtz_dp_result_t*)hx_realloc(D->rs, sizeof(wtz_dp_result_t) * step); }
		uint64_t need = 0;
		for(size_t i = 0; i < step; i++){
			const wx_job_t *j = &jobs[at + i]; wtz_dp_problem_t *p = &D->pr[i];
			p->q_read = j->q_read; p->t_read = j->t_read; p->q_rev = j->q_rev; p->t_rev = j->t_rev; p->q_from = j->q_from; p->t_from = j->t_from;
			p->q_strand = p->t_strand = j->strand; p->q_len = j->q_len; p->t_len = j->t_len; p->init_score = j->init_score; p->W = D->W;
			need += (uint64_t)(j->q_len > 0 ? j->q_len : 0) + (uint64_t)(j->t_len > 0 ? j->t_len : 0) + 2;      /* a CIGAR has at most one operation per base of either side */
		}
		if(need > D->capcig){ D->capcig = need + need / 4; D->cig = (uint32_t*)hx_realloc(D->cig, 4 * D->capcig); }
		const int rc = wtz_extend_batch(D->ctx, D->pr, (uint32_t)step, D->rs, D->cig, D->capcig);
		if(rc == WTZ_E_POOL){
			if(step == 1){ fprintf(stderr, " -- scratch pool too small even for one extension: %s --\n", wtz_last_error()); WX_DIE(); }
			step = (step + 1) / 2; continue;
		}
		DIE_WTZ(rc, "wtz_extend_batch");
		for(size_t i = 0; i < step; i++){
			const wtz_dp_result_t *r = &D->rs[i]; wx_jobres_t *o = &res[at + i];
			o->x.score = r->score; o->x.tb = r->tb; o->x.te = r->te; o->x.qb = r->qb; o->x.qe = r->qe; o->x.aln = r->aln; o->x.mat = r->mat; o->x.mis = r->mis; o->x.ins = r->ins; o->x.del = r->del;
			uint32_t *dst = wx_cigar_space(W, r->cigar_len);
			memcpy(dst, D->cig + r->cigar_off, 4 * (size_t)r->cigar_len);
			o->cig_off = (uint64_t)(dst - W->cigar_pool); o->cig_len = r->cigar_len;
		}
		at += step;
	}
}

int main(int argc, char **argv){
	wx_t *W = (wx_t*)calloc(1, sizeof(wx_t));
	if(wx_parse_args(&W->O, argc, argv)) return wx_usage();
	wx_opt_t *o = &W->O;
	if(wtz_device_count() <= 0){ fprintf(stderr, " -- no HIP device visible: wtext (MI355X build) has no CPU path for the extension: %s --\n", wtz_last_error()); return 1; }
	wtz_params_c P; memset(&P, 0, sizeof P);
	P.ksize = 16; P.zsize = 10; P.hk = 1; P.hz = 1; P.ksave = 4; P.kovl = 300; P.ncand = 500; P.nbest = 100; P.kwin = 800; P.kstep = 400; P.ztot = 300; P.zovl = 200;
	P.max_zmer_freq = 64; P.max_kmer_var = 2; P.win_rep_norm = 20; P.win_rep_cutoff = 100;
	P.w = 50; P.ew = 800; P.W = 3200; P.M = o->M; P.X = o->X; P.O = o->O; P.E = o->E; P.T = o->T; P.min_score = 200; P.min_id = 0.5f;
	P.xvar = 128; P.yvar = 64; P.min_block_len = 160; P.max_overhang = 256; P.deviation_penalty = 1.0f; P.gap_penalty = 0.05f;
	static wx_ctxjob_t cj; cj.P = &P; cj.device = o->gpu; cj.pool_bytes = (o->pool_gb ? o->pool_gb : 16) << 30; cj.ctx = NULL; cj.rc = WTZ_OK;
	const int cj_started = (pthread_create(&cj.th, NULL, wx_ctxjob_main, &cj) == 0);      /* the pool's hipMalloc runs while the reads are loaded */
	W->st.keep_text = 1;
	wx_load_inputs(W);
	const uint32_t n = W->n_pb;
	uint64_t *rdoff = (uint64_t*)hx_realloc(NULL, 8 * ((size_t)n + 1));
	for(uint32_t i = 0; i < n; i++) rdoff[i] = W->st.reads[i].off;
	if(cj_started) pthread_join(cj.th, NULL); else wx_ctxjob_main(&cj);
	if(cj.rc != WTZ_OK){ fprintf(stderr, " -- wtz_ctx_create failed: %s --\n", cj.err); WX_DIE(); }
	wx_dev_t D; memset(&D, 0, sizeof D); D.ctx = cj.ctx; D.W = o->W;
	/* the bases travel as text and are packed on the device (wtz_upload_reads_ascii); the host needs the packed bank too - the re-scoring of the
	 * clipped CIGARs compares bases (wtext.c:237-240) - and fetches it back once */
	int rc = wtz_upload_reads_ascii(D.ctx, W->st.text, W->st.nbase, rdoff, W->pblen, n, 0, NULL); DIE_WTZ(rc, "wtz_upload_reads_ascii");
	{ const uint64_t nw = (W->st.nbase + 31) / 32;
	  W->st.bits = (uint64_t*)hx_realloc(NULL, 8 * (nw + 2)); W->st.capw = nw + 2; W->st.bits[nw] = W->st.bits[nw + 1] = 0;
	  rc = wtz_fetch_read_bits(D.ctx, W->st.bits, nw); DIE_WTZ(rc, "wtz_fetch_read_bits"); }
	free(W->st.text); W->st.text = NULL; free(rdoff);
	W->backend = &D;
	rc = wx_run(W);
	{ wtz_counters_t c; if(wtz_get_counters(D.ctx, &c) == WTZ_OK) fprintf(stderr, "[wtext-mi355x] extension kernels %.1f ms, %llu DP cells (K-sw3)\n", c.ms_stitch, (unsigned long long)c.cells_shift); }
	wtz_ctx_destroy(D.ctx);
	return rc;
}
