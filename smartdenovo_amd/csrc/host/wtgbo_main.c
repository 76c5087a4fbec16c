/*
 * wtgbo_main.c — drop-in `wtgbo`: the graph-based overlapper that follows wtzmo in the zmo pipeline
 * (smartdenovo.pl:60-61: `wtgbo -t N -i reads.fa.gz -j zmo.ovl.short -fo -`).  Same options, same 17-column records and -9 pair
 * file as the reference's wtgbo.c; the output is that of `wtgbo -t 1`.
 *
 * Host (this file + wtgbo_core.h + wtgbo_graph.h, plain C): inputs, the overlap graph, the candidate walks, the commit order.
 * Device (libwtzmo_hip.so through include/wtzmo_hip.h): every pair alignment — align_hzmaux (hzm_aln.h:1684-1775) is the pair
 * pipeline of wtzmo's zmo engine with three differences, all carried by wtz_params_c.aux_strand = 1 and by how the reads are uploaded:
 *   1. the candidate read is reverse-complemented BEFORE its z-mers are taken when its strand is '-' (wtgbo.c:48-49), and a z-mer's
 *      span under homopolymer compression is not mirror-symmetric (it ends on the first base of its last run).  So every read is uploaded
 *      twice: read i and, as read n + i, its reverse complement; a '-' job aligns read n + qry;
 *   2. only same-strand matches survive (filter_by_region_hzmps dir 0, hzm_aln.h:1695), filtered before the window merge;
 *   3. no n_hits gate, chain threshold -R, window step 0, -Z 100.
 * The gates behind the stitched alignment (hzm_aln.h:1715-1718) are integer / float compares on the returned kswx_t and run at commit.
 * There is no CPU implementation of the alignment in this program: without a HIP device it exits with an error.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <pthread.h>
#include "wtgbo_core.h"

#define DIE_WTZ(rc, what) do { if((rc) != WTZ_OK){ fprintf(stderr, " -- %s failed: %s --\n", what, wtz_last_error()); fflush(NULL); _exit(1); } } while(0)

typedef struct {
	wtz_ctx_t *ctx; uint32_t n_rd; int zindex_all;
	uint32_t *pq, *pc, *itp, *ids; uint8_t *itd; wtz_pair_summary_t *sum; wtz_aln_result_t *aln; uint32_t cap;
	uint32_t *cig; uint64_t capcig;
	uint8_t *mark;            /* per uploaded read: in the current z-index subset */
} gbo_dev_t;

typedef struct { const wtz_params_c *P; int device; uint64_t pool_bytes; wtz_ctx_t *ctx; int rc; char err[256]; pthread_t th; } gbo_ctxjob_t;
static void *gbo_ctxjob_main(void *arg){
	gbo_ctxjob_t *j = (gbo_ctxjob_t*)arg;
	j->rc = wtz_ctx_create(j->device, j->P, j->pool_bytes, &j->ctx);
	if(j->rc != WTZ_OK) snprintf(j->err, sizeof j->err, "%s", wtz_last_error());
	return NULL;
}

/* 0 = done, 1 = the scratch pool was too small for this many pairs */
static int gbo_align_range(gbo_t *G, gbo_dev_t *D, const gbo_job_t *jobs, uint32_t n, gbo_res_t *res){
	int rc;
	if(n > D->cap){
		D->cap = n;
		D->pq = (uint32_t*)hx_realloc(D->pq, 4 * (size_t)n); D->pc = (uint32_t*)hx_realloc(D->pc, 4 * (size_t)n);
		D->itp = (uint32_t*)hx_realloc(D->itp, 4 * (size_t)n); D->itd = (uint8_t*)hx_realloc(D->itd, (size_t)n);
		D->ids = (uint32_t*)hx_realloc(D->ids, 8 * (size_t)n);
		D->sum = (wtz_pair_summary_t*)hx_realloc(D->sum, sizeof(wtz_pair_summary_t) * (size_t)n);
		D->aln = (wtz_aln_result_t*)hx_realloc(D->aln, sizeof(wtz_aln_result_t) * (size_t)n);
	}
	for(uint32_t i = 0; i < n; i++){ D->pq[i] = jobs[i].obj; D->pc[i] = jobs[i].qry + (jobs[i].dir ? D->n_rd : 0); res[i].ok = 0; res[i].cig_len = 0; res[i].cig_off = 0; }
	if(!D->zindex_all){
		/* z-mer index of the reads of this batch only (ascending ids) */
		uint32_t k = 0;
		for(uint32_t i = 0; i < n; i++){ if(!D->mark[D->pq[i]]){ D->mark[D->pq[i]] = 1; D->ids[k++] = D->pq[i]; } if(!D->mark[D->pc[i]]){ D->mark[D->pc[i]] = 1; D->ids[k++] = D->pc[i]; } }
		for(uint32_t i = 0; i < k; i++) D->mark[D->ids[i]] = 0;
		/* ascending order */
		for(uint32_t gap = k / 2; gap > 0; gap /= 2) for(uint32_t i = gap; i < k; i++){ uint32_t v = D->ids[i], j = i; while(j >= gap && D->ids[j - gap] > v){ D->ids[j] = D->ids[j - gap]; j -= gap; } D->ids[j] = v; }
		rc = wtz_zindex_build_subset(D->ctx, D->ids, k); if(rc == WTZ_E_POOL) return 1; DIE_WTZ(rc, "wtz_zindex_build_subset");
	}
	rc = wtz_batch_begin(D->ctx); DIE_WTZ(rc, "wtz_batch_begin");
	rc = wtz_pairs_seed(D->ctx, D->pq, D->pc, n, D->sum); if(rc == WTZ_E_POOL) return 1; DIE_WTZ(rc, "wtz_pairs_seed");
	uint32_t m = 0;
	for(uint32_t i = 0; i < n; i++) if(D->sum[i].gate && D->sum[i].nwin[0]){ D->itp[m] = i; D->itd[m] = 0; m++; }      /* hzm_aln.h:1698-1699: windows, chain >= -R */
	if(m == 0) return 0;
	rc = wtz_pairs_align(D->ctx, D->itp, D->itd, m, D->aln); if(rc == WTZ_E_POOL) return 1; DIE_WTZ(rc, "wtz_pairs_align");
	uint64_t tot = 0; for(uint32_t k = 0; k < m; k++) tot += D->aln[k].cigar_len;
	if(tot + 1 > D->capcig){ D->capcig = tot + 1 + tot / 2; D->cig = (uint32_t*)hx_realloc(D->cig, 4 * D->capcig); }
	rc = wtz_fetch_cigars(D->ctx, D->cig, tot); if(rc == WTZ_E_POOL) return 1; DIE_WTZ(rc, "wtz_fetch_cigars");
	uint32_t *dst = gbo_cigar_space(G, tot);
	memcpy(dst, D->cig, 4 * (size_t)tot);
	const uint64_t base = (uint64_t)(dst - G->cigar_pool);
	uint64_t off = 0;
	for(uint32_t k = 0; k < m; k++){
		const wtz_aln_result_t *a = &D->aln[k]; gbo_res_t *r = &res[D->itp[k]];
		r->ok = a->n_regs > 0;                                   /* hzm_aln.h:1712 */
		r->score = a->score; r->tb = a->tb; r->te = a->te; r->qb = a->qb; r->qe = a->qe; r->aln = a->aln; r->mat = a->mat; r->mis = a->mis; r->ins = a->ins; r->del = a->del;
		r->cig_off = base + off; r->cig_len = a->cigar_len; off += a->cigar_len;
	}
	return 0;
}

static void gbo_align_jobs(gbo_t *G, const gbo_job_t *jobs, size_t n, gbo_res_t *res){
	gbo_dev_t *D = (gbo_dev_t*)G->backend;
	size_t at = 0, step = n;
	while(at < n){
		if(step > n - at) step = n - at;
		const uint64_t ncig0 = G->ncig;
		if(gbo_align_range(G, D, jobs + at, (uint32_t)step, res + at)){
			G->ncig = ncig0;
			if(step == 1){ fprintf(stderr, " -- scratch pool too small even for one pair: %s --\n", wtz_last_error()); fflush(NULL); _exit(1); }
			step = (step + 1) / 2;
			continue;
		}
		at += step;
	}
}

int main(int argc, char **argv){
	gbo_t *G = (gbo_t*)calloc(1, sizeof(gbo_t));
	if(gbo_parse_args(&G->O, argc, argv)) return gbo_usage();
	gbo_opt_t *o = &G->O;
	if(o->zsize < 5 || o->zsize > 16){ fprintf(stderr, " -- -z must be within 5..16 --\n"); return gbo_usage(); }
	if(wtz_device_count() <= 0){ fprintf(stderr, " -- no HIP device visible: wtgbo (MI355X build) has no CPU path for the alignment: %s --\n", wtz_last_error()); return 1; }
	/* the context - above all the hipMalloc of its scratch pool, ~35 ms per GB - is created on a helper thread while the reads are loaded.  Default pool 16 GB, not the
	 * library's 45 % of the HBM: a pass aligns thousands of pairs, not hundreds of thousands (E. coli shape: 5.4 s wall with the 128 GB default, 1.0 s with 16 GB) */
	wtz_params_c P; memset(&P, 0, sizeof P);
	P.ksize = 16; P.zsize = (uint32_t)o->zsize; P.hk = 1; P.hz = (uint32_t)o->hz; P.ksave = 4; P.kovl = 300; P.ncand = 500; P.nbest = 100;
	P.kwin = (uint32_t)o->kwin; P.kstep = (uint32_t)o->kstep; P.ztot = (uint32_t)o->zovl; P.zovl = (uint32_t)o->zovl;
	P.max_kmer_freq = 0; P.max_zmer_freq = (uint32_t)o->zcut; P.max_kmer_var = (uint32_t)o->kvar;
	P.win_rep_norm = 20; P.win_rep_cutoff = 100;
	P.w = o->w; P.ew = o->ew; P.W = o->W; P.M = o->M; P.X = o->X; P.O = o->O; P.E = o->E; P.T = o->T;
	P.min_score = o->min_score; P.min_id = o->min_id;
	P.dot_matrix = 0; P.xvar = 128; P.yvar = 64; P.min_block_len = 160; P.max_overhang = 256; P.deviation_penalty = 1.0f; P.gap_penalty = 0.05f;
	P.refine = o->refine; P.aux_strand = 1;      /* -n: the device applies the gates of hzm_aln.h:1715-1718 before it refines (wtz_task_refine) */
	static gbo_ctxjob_t cj; cj.P = &P; cj.device = o->gpu; cj.pool_bytes = (o->pool_gb ? o->pool_gb : 16) << 30; cj.ctx = NULL; cj.rc = WTZ_OK;
	const int cj_started = (pthread_create(&cj.th, NULL, gbo_ctxjob_main, &cj) == 0);
	G->st.keep_text = (o->ingest_host == 0);
	gbo_load_inputs(G);            /* its error paths leave through exit(): join first if that ever matters - they fire within milliseconds of the start */
	/* ---- device: every read and its reverse complement.  The bases travel as text and are packed on the device (wtz_upload_reads_ascii),
	 * the reverse-complement views are made there too (wtz_append_revcomp_views): the host never touches a base ---- */
	const uint32_t n = G->n_rd;
	uint64_t *rdoff = (uint64_t*)hx_realloc(NULL, 8 * ((size_t)n + 1));
	uint64_t tot = 0;
	for(uint32_t i = 0; i < n; i++){ rdoff[i] = G->st.reads[i].off; tot += G->st.reads[i].len; }
	gbo_dev_t D; memset(&D, 0, sizeof D);
	D.n_rd = n;
	if(cj_started) pthread_join(cj.th, NULL); else gbo_ctxjob_main(&cj);
	if(cj.rc != WTZ_OK){ fprintf(stderr, " -- wtz_ctx_create failed: %s --\n", cj.err); fflush(NULL); _exit(1); }
	D.ctx = cj.ctx;
	int rc;
	if(G->st.keep_text){ rc = wtz_upload_reads_ascii(D.ctx, G->st.text, G->st.nbase, rdoff, G->rdlen, n, 0, NULL); DIE_WTZ(rc, "wtz_upload_reads_ascii"); free(G->st.text); G->st.text = NULL; }
	else { rc = wtz_upload_reads(D.ctx, G->st.bits, (G->st.nbase + 31) >> 5, rdoff, G->rdlen, n); DIE_WTZ(rc, "wtz_upload_reads"); }
	rc = wtz_append_revcomp_views(D.ctx); DIE_WTZ(rc, "wtz_append_revcomp_views");
	free(rdoff);
	D.zindex_all = o->zindex_batch < 0 ? (2 * tot <= 2400000000ull) : !o->zindex_batch;      /* 16 B per indexed base: all reads while that stays under ~40 GB */
	if(D.zindex_all){ rc = wtz_zindex_build(D.ctx); DIE_WTZ(rc, "wtz_zindex_build"); }
	else D.mark = (uint8_t*)calloc((size_t)n * 2 + 1, 1);
	fprintf(stderr, "[wtgbo-mi355x] %u reads, %llu bp (+ reverse complements) on device %d, z-mer index %s\n", n, (unsigned long long)tot, o->gpu, D.zindex_all ? "of all reads" : "per batch");
	G->backend = &D;
	rc = gbo_run(G);
	{ wtz_counters_t c; if(wtz_get_counters(D.ctx, &c) == WTZ_OK) fprintf(stderr, "[wtgbo-mi355x] kernel ms: zindex %.1f pairs %.1f winalign %.1f stitch %.1f; DP cells K-sw1 %llu K-sw2 %llu K-sw3 %llu\n",
			c.ms_zindex, c.ms_pairs, c.ms_winalign, c.ms_stitch, (unsigned long long)c.cells_fixed, (unsigned long long)c.cells_global, (unsigned long long)c.cells_shift); }
	wtz_ctx_destroy(D.ctx);
	return rc;
}
