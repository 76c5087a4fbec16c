/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see ora_util.h).
 *
 * ora_overlap.h — the per-query worker, the strictly sequential commit loop that defines
 * `wtzmo -t 1` semantics, and the .ovl record writer.
 * Restates:
 *   - thread_mzmo_func (one query)   reference wtzmo.c:722-1167
 *   - print_hits_wtzmo               reference wtzmo.c:1170-1249
 *   - overlap_wtzmo dispatch order   reference wtzmo.c:1251-1357 with thread.h:63-197 at one
 *                                    worker: the `masked` test of query j (1315) runs BEFORE
 *                                    the results of the previously dispatched query are
 *                                    merged (1316-1329), so masking lags by one query.
 */
#ifndef ORA_OVERLAP_H
#define ORA_OVERLAP_H

#include "ora_index.h"
#include "ora_align.h"
#include "ora_refine.h"
#include "ora_dotmatrix.h"

typedef struct {
	/* seeding */
	uint32_t ksize, zsize, hk, hz, ksave, kovl, ncand, nbest, n_idx;
	uint32_t kwin, kstep, ztot, zovl, max_kmer_freq, max_zmer_freq, max_kmer_var;
	float win_rep_norm, win_rep_cutoff;
	/* alignment */
	int w, ew, W, M, X, O, E, T;
	int min_score; float min_id;
	uint32_t max_unalign_in_contained, max_unalign_in_dovetail;
	/* dot-matrix engine */
	int dot_matrix, xvar, yvar, min_block_len, max_overhang;
	float deviation_penalty, gap_penalty;
	int do_align, refine;
} ora_params_t;

static void ora_params_default(ora_params_t *p){      /* wtzmo.c:1543-1588 + 174-175 */
	memset(p, 0, sizeof *p);
	p->w = 50; p->ew = 800; p->W = 3200; p->M = 2; p->X = -5; p->O = -3; p->E = -1; p->T = -50;
	p->min_score = 200; p->min_id = 0.5f; p->hk = 1; p->hz = 1; p->ksize = 16; p->zsize = 10;
	p->kwin = 800; p->kstep = 400; p->kovl = 300; p->ksave = 4; p->n_idx = 1;
	p->win_rep_norm = 20; p->win_rep_cutoff = 100; p->ncand = 500; p->nbest = 100;
	p->ztot = 300; p->zovl = 200; p->max_kmer_freq = 0; p->max_zmer_freq = 64; p->max_kmer_var = 2;
	p->dot_matrix = 0; p->xvar = 128; p->yvar = 64; p->min_block_len = 160; p->max_overhang = 256;
	p->deviation_penalty = 1.0f; p->gap_penalty = 0.05f;
	p->max_unalign_in_contained = 0; p->max_unalign_in_dovetail = 200;
	p->do_align = 1; p->refine = 0;
}

/* ---- closed pair set (u64hash in the reference; only membership is observable here) ---- */
typedef struct { uint64_t *tab; size_t cap, n; } ora_u64set_t;
static inline uint64_t ora_mix64(uint64_t x){ x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
static void ora_u64set_put(ora_u64set_t *s, uint64_t v);
static void ora_u64set_grow(ora_u64set_t *s){
	size_t oc = s->cap; uint64_t *ot = s->tab;
	s->cap = oc ? oc * 2 : 1024; s->n = 0;
	s->tab = (uint64_t*)ora_xrealloc(NULL, s->cap * 8); memset(s->tab, 0xFF, s->cap * 8);
	for(size_t i = 0; i < oc; i++) if(ot[i] != ~0ULL) ora_u64set_put(s, ot[i]);
	free(ot);
}
static void ora_u64set_put(ora_u64set_t *s, uint64_t v){
	if((s->n + 1) * 2 > s->cap) ora_u64set_grow(s);
	size_t m = s->cap - 1, i = ora_mix64(v) & m;
	while(s->tab[i] != ~0ULL){ if(s->tab[i] == v) return; i = (i + 1) & m; }
	s->tab[i] = v; s->n++;
}
static int ora_u64set_has(const ora_u64set_t *s, uint64_t v){
	if(s->cap == 0) return 0;
	size_t m = s->cap - 1, i = ora_mix64(v) & m;
	while(s->tab[i] != ~0ULL){ if(s->tab[i] == v) return 1; i = (i + 1) & m; }
	return 0;
}
static inline uint64_t ora_pair_key(uint64_t a, uint64_t b){     /* ovl_uniq_long_id(.,.,0), wtzmo.c:83-84 */
	return a < b ? ((a << 33) | (b << 1)) : ((b << 33) | (a << 1));
}

typedef struct {
	uint32_t pb1, pb2, dir2;
	int qb, qe, tb, te, score, mat, mis, ins, del, aln;
	char *cigar;
} ora_hit_t;
ORA_VEC(vec_hit, ora_hit_t)

typedef struct {
	ora_store_t   st;
	uint32_t      n_qr;
	ora_kindex_t  ix;
	ora_params_t  P;
	uint8_t      *masked;
	uint32_t     *rdcovs;
	ora_u64set_t  closed;
	vec_u64      *rdhits;      /* per-read candidate heaps when n_idx > 1 */
	uint64_t      pair_bp;     /* metric numerator: sum len(a)+len(b) over pairs entering alignment */
	uint64_t      n_pairs;
} ora_ctx_t;

/* per-worker state: results of one query + scratch */
typedef struct {
	vec_hit hits; vec_u32 masks; vec_u64 closed; vec_win seeds;    /* results */
	uint32_t rd_id, bcov;
	/* scratch */
	vec_kgroup groups; vec_khit khits; vec_u64 cand;
	ora_ztable_t zt; vec_u8 kcnts; vec_zhit cache, anchors, anchors2;
	vec_win windows, windows2; ora_winscratch_t wsc; vec_i32 chainmem;
	vec_u16 windeps; vec_f32 weights;
	vec_u8 pb1, pb2; vec_u32 cigar_cache, cigars, tmp_cigar; vec_reg regs; ora_swmem_t swmem; ora_refmem_t refmem;
	vec_u8 text; vec_u32 maskset;
	ora_dm_scratch_t dm;
	ora_win_t SEED[2];
} ora_worker_t;

static inline void ora_maskset_put(vec_u32 *s, uint32_t v){ for(size_t i = 0; i < s->n; i++) if(s->a[i] == v) return; vec_u32_push(s, v); }

#define ORA_SEED_GT(a, b) ((b).ovl > (a).ovl)
ORA_DEFINE_SORT(ora_sort_seeds_desc, ora_win_t, ORA_SEED_GT)

/* wtzmo.c:806-822: candidate list of one query after the closed-pair filter */
static void ora_query_candidates(ora_ctx_t *C, ora_worker_t *K, uint32_t pbid, vec_u64 *cand){
	ora_kparams_t kp; kp.ksize = C->P.ksize; kp.hk = C->P.hk; kp.ksave = C->P.ksave; kp.kovl = C->P.kovl; kp.ncand = C->P.ncand; kp.max_kmer_freq = C->P.max_kmer_freq;
	ora_query_groups(&C->st, &C->ix, &kp, pbid, &K->groups, &K->khits);
	ora_candidates_from_groups(K->groups.a, K->groups.n, C->P.kovl, C->P.ncand, cand);
	for(size_t i = 0; i < cand->n; i++){
		uint32_t id2 = (uint32_t)(cand->a[i] >> 32);
		if(ora_u64set_has(&C->closed, ora_pair_key(pbid, id2))) cand->a[i] &= 0xFFFFFFFF00000000ULL;
	}
	ora_sort_cand_desc(cand->a, cand->n, NULL);
	while(cand->n && (cand->a[cand->n - 1] & 0xFFFFFFFFu) == 0) cand->n--;
}

static void ora_worker_run(ora_ctx_t *C, ora_worker_t *K, uint32_t pbid, uint32_t bcov_in, int just_query){
	const ora_params_t *P = &C->P;
	const ora_read_t *reads = C->st.reads.a;
	uint32_t nbest, i, j, k, dir, id2, ncand, ol;
	int alen, blen;
	K->rd_id = pbid; K->bcov = bcov_in;
	nbest = (uint32_t)(((size_t)P->nbest) * reads[pbid].len / C->ix.avg_rdlen);
	if(nbest < P->nbest) nbest = P->nbest;
	if(K->bcov >= nbest) return;
	alen = (int)reads[pbid].len;
	vec_u64 *cand;
	if(C->rdhits) cand = &C->rdhits[pbid]; else { K->cand.n = 0; cand = &K->cand; }
	ora_query_candidates(C, K, pbid, cand);
	if(just_query) return;
	K->seeds.n = 0; K->windows.n = 0; K->anchors.n = 0; K->cache.n = 0;
	for(size_t h = 0; h < K->hits.n; h++) free(K->hits.a[h].cigar);
	K->hits.n = 0;
	vec_u16_reserve(&K->windeps, (size_t)alen + 1); memset(K->windeps.a, 0, (size_t)alen * 2);
	vec_f32_reserve(&K->weights, (size_t)alen + 1);
	vec_u8_reserve(&K->pb1, (size_t)alen + 8);
	ora_unpack(&C->st, reads[pbid].off, (uint32_t)alen, 0, K->pb1.a);
	ora_ztable_build(&K->zt, K->pb1.a, (uint32_t)alen, P->zsize, (int)P->hz, P->max_zmer_freq);
	for(i = 0; i < cand->n; i++){
		id2 = (uint32_t)(cand->a[i] >> 32);
		blen = (int)reads[id2].len;
		vec_u8_reserve(&K->pb2, (size_t)blen + 8);
		ora_unpack(&C->st, reads[id2].off, (uint32_t)blen, 0, K->pb2.a);
		ora_zmatch(&K->zt, K->pb2.a, (uint32_t)blen, P->zsize, (int)P->hz, P->max_zmer_freq, P->max_kmer_var, &K->kcnts, &K->cache);
		if(K->cache.n * P->zsize < P->ztot) continue;
		if(P->dot_matrix){
			vec_u64_push(&K->closed, ora_pair_key(id2, pbid));
			ora_dm_result_t r = ora_dot_matrix_align(&K->cache, &K->dm, alen, blen, P->xvar, P->yvar, P->min_block_len, P->max_overhang, P->deviation_penalty, P->gap_penalty);
			ol = (uint32_t)ORA_MAX(r.qe - r.qb, r.te - r.tb);
			if(r.score >= P->min_score && r.score >= (int)(P->min_id * ol)){
				ora_hit_t H; memset(&H, 0, sizeof H);
				H.pb1 = pbid; H.pb2 = id2; H.dir2 = (uint32_t)r.dir; H.score = r.score;
				H.tb = r.tb; H.te = r.te; H.qb = r.qb; H.qe = r.qe; H.mat = r.score; H.aln = (int)ol; H.cigar = NULL;
				vec_hit_push(&K->hits, H);
			}
			continue;
		}
		vec_zhit_reserve(&K->cache, K->cache.n + 1); memset(&K->cache.a[K->cache.n], 0, sizeof(ora_zhit_t));
		ora_sort_zhit_off12(K->cache.a, K->cache.n, NULL);
		for(dir = 0; dir < 2; dir++){
			ora_win_t *S = &K->SEED[dir];
			S->pb2 = id2; S->dir = dir; S->ovl = 0; S->closed = 1;
			K->windows2.n = 0; K->anchors2.n = 0;
			if(ora_merge_windows(K->cache.a, (uint32_t)K->cache.n, (int)dir, &K->windows2, &K->anchors2, &K->wsc, P->zsize, P->kwin, P->kstep, P->zovl) == 0) continue;
			S->ovl = ORA_OVL29(ora_chain_windows(K->windows2.a, 0, (uint32_t)K->windows2.n, P->W, &K->chainmem));
			if(S->ovl < P->ztot) continue;
			S->anchors[0] = (uint32_t)K->windows.n;
			for(j = 0; j < K->windows2.n; j++){
				ora_win_t *zp = &K->windows2.a[j];
				if(zp->closed) continue;
				vec_zhit_append(&K->anchors, K->anchors2.a + zp->anchors[0], zp->anchors[1] - zp->anchors[0]);
				zp->anchors[1] = zp->anchors[1] - zp->anchors[0];
				zp->anchors[0] = (uint32_t)K->anchors.n - zp->anchors[1];
				zp->anchors[1] = (uint32_t)K->anchors.n;
				vec_win_push(&K->windows, *zp);
				for(k = (uint32_t)zp->beg[0]; (int)k < zp->end[0]; k++) K->windeps.a[k]++;
			}
			S->anchors[1] = (uint32_t)K->windows.n;
			S->closed = 0;
		}
		dir = (K->SEED[0].ovl < K->SEED[1].ovl);
		if(K->SEED[dir].ovl >= P->ztot) vec_win_push(&K->seeds, K->SEED[dir]);
	}
	if(P->dot_matrix) return;
	/* repeat weighting (wtzmo.c:933-936): float/double mix kept as written */
	for(i = 0; (int)i < alen; i++)
		K->weights.a[i] = (K->windeps.a[i] <= P->win_rep_norm) ? 1.0
			: ((K->windeps.a[i] >= P->win_rep_cutoff) ? 0.0 : P->win_rep_norm / (float)K->windeps.a[i]);
	for(i = 0; (int)i < alen; i++)
		K->weights.a[i] = K->weights.a[i] * (0.3 + 0.7 * (ORA_ABSDIFF(((int)i), alen / 2) / (alen / 2.0)));
	K->maskset.n = 0;
	for(i = 0; i < K->seeds.n; i++){
		ora_win_t *seed = &K->seeds.a[i];
		double avg;
		blen = (int)reads[seed->pb2].len;
		if(seed->closed) continue;
		ol = 0;
		for(j = seed->anchors[0]; j < seed->anchors[1]; j++){
			const ora_win_t *zp = &K->windows.a[j];
			if(zp->closed) continue;
			avg = (zp->end[0] - zp->beg[0]) * K->weights.a[(zp->beg[0] + zp->end[0]) / 2];
			avg = avg * (0.3 + 0.7 * (ORA_ABSDIFF(((int)((zp->beg[1] + zp->end[1]) / 2)), blen / 2) / (blen / 2.0)));
			ol += avg;
		}
		seed->ovl = ORA_OVL29(ol);
		if(ol * P->win_rep_cutoff < P->ztot * P->win_rep_norm) seed->closed = 1;
	}
	ora_sort_seeds_desc(K->seeds.a, K->seeds.n, NULL);
	if(!P->do_align) return;
	ncand = P->ncand;
	for(i = 0; i < K->seeds.n && i < ncand; i++){
		ora_win_t *seed = &K->seeds.a[i];
		if(seed->closed){ ncand++; continue; }
		vec_u64_push(&K->closed, ora_pair_key(seed->pb2, pbid));
		blen = (int)reads[seed->pb2].len;
		vec_u8_reserve(&K->pb2, (size_t)blen + 8);
		ora_unpack(&C->st, reads[seed->pb2].off, (uint32_t)blen, (int)seed->dir, K->pb2.a);
		K->cigar_cache.n = 0; K->regs.n = 0;
		for(j = seed->anchors[0]; j < seed->anchors[1]; j++){
			const ora_win_t *zp = &K->windows.a[j];
			if(zp->closed) continue;
			ora_reg_t R;
			R.cigar_off = (uint32_t)K->cigar_cache.n;
			R.x = ora_align_window(K->pb1.a, K->pb2.a, zp, K->anchors.a, &K->cigar_cache, &K->swmem, &K->tmp_cigar, P->w, P->M, P->X, P->O, P->O, P->E, P->T);
			R.cigar_len = (uint32_t)K->cigar_cache.n - R.cigar_off;
			vec_u32_push(&K->cigar_cache, 0x0F);
			if(R.x.aln * 2 < (int)P->zovl || R.x.mat < R.x.aln * P->min_id) continue;
			vec_reg_push(&K->regs, R);
		}
		if(K->regs.n == 0){ seed->closed = 1; ncand++; continue; }
		int esti[2] = {0, alen};
		ora_aln_t x = ora_stitch_windows(alen, blen, K->regs.a, K->regs.n, esti, K->pb1.a, K->pb2.a, K->cigar_cache.a, &K->cigars, &K->swmem, &K->tmp_cigar,
			P->W, P->ew, P->w, P->M, P->X, P->O, P->O, P->E, P->T);
		K->regs.n = 0;
		if(P->refine){       /* wtzmo.c:1031-1034: the stitched CIGAR becomes the guide of the re-alignment */
			K->cigar_cache.n = 0; vec_u32_append(&K->cigar_cache, K->cigars.a, K->cigars.n);
			x = ora_refine_alignment(K->pb2.a, x.qb, K->pb1.a, x.tb, (int)P->w, P->M, P->X, P->O, P->O, P->E, K->cigar_cache.a, K->cigar_cache.n, &K->refmem, &K->cigars);
		}
		if(x.score < P->min_score || x.mat < x.aln * P->min_id) continue;
		ora_hit_t H; memset(&H, 0, sizeof H);
		K->text.n = 0; ora_cigar_text(&K->text, K->cigars.a, K->cigars.n); vec_u8_push(&K->text, 0);
		H.pb1 = pbid; H.pb2 = seed->pb2; H.dir2 = seed->dir; H.score = x.score;
		H.tb = x.tb; H.te = x.te; H.qb = x.qb; H.qe = x.qe;
		H.mat = x.mat; H.mis = x.mis; H.ins = x.ins; H.del = x.del; H.aln = x.aln;
		H.cigar = strdup((char*)K->text.a);
		vec_hit_push(&K->hits, H);
		{
			const ora_hit_t *hit = &H;
			uint32_t len1 = reads[hit->pb1].len, len2 = reads[hit->pb2].len;
			uint32_t x1, x2, x3, x4;
			x1 = (uint32_t)ORA_MIN(hit->tb, hit->qb);
			x2 = (uint32_t)ORA_MIN(((int)len1) - hit->te, ((int)len2) - hit->qe);
			if(x1 + x2 <= P->max_unalign_in_dovetail){
				x3 = ((hit->tb == 0 && hit->qb) || (hit->te == (int)len1 && hit->qe < (int)len2));
				x4 = ((hit->qb == 0 && hit->tb) || (hit->qe == (int)len2 && hit->te < (int)len1));
				x1 = len2 + (uint32_t)hit->qb - (uint32_t)hit->qe;
				x2 = len1 + (uint32_t)hit->tb - (uint32_t)hit->te;
				if(x1 <= P->max_unalign_in_contained && x3 == 0){
					if(x2 <= P->max_unalign_in_contained && x4 == 0){
						if(len1 > len2){ ora_maskset_put(&K->maskset, hit->pb2); }
						else if(len1 < len2){ ora_maskset_put(&K->maskset, hit->pb1); break; }
						else if(hit->pb2 > hit->pb1){ ora_maskset_put(&K->maskset, hit->pb2); continue; }
						else { ora_maskset_put(&K->maskset, hit->pb1); break; }
					} else { ora_maskset_put(&K->maskset, hit->pb2); continue; }
					ncand++;
				} else if(x2 <= P->max_unalign_in_contained && x4 == 0){
					ora_maskset_put(&K->maskset, hit->pb1); break;
				}
				K->bcov++;
				if(K->bcov >= nbest) break;
			}
		}
	}
	for(i = 0; i < K->maskset.n; i++) vec_u32_push(&K->masks, K->maskset.a[i]);
}

/* wtzmo.c:1170-1249 */
static uint64_t ora_flush_hits(ora_ctx_t *C, ora_worker_t *K, FILE *out){
	const ora_read_t *reads = C->st.reads.a;
	uint64_t ret = 0;
	if(!C->P.do_align){
		for(size_t i = 0; i < K->seeds.n; i++){
			const ora_win_t *seed = &K->seeds.a[i];
			if(seed->closed) continue;
			fprintf(out, "# %s\t%c\t%d", reads[K->rd_id].name, '+', reads[K->rd_id].len);
			fprintf(out, "\t%s\t%c\t%d", reads[seed->pb2].name, "+-"[seed->dir], reads[seed->pb2].len);
			fprintf(out, "\t%d\n", seed->ovl);
		}
		K->seeds.n = 0;
	} else {
		for(size_t j = 0; j < K->hits.n; j++){
			ora_hit_t *hit = &K->hits.a[j];
			ret++;
			if(hit->aln == 0) hit->aln = 1;
			uint32_t x1 = (uint32_t)ORA_MIN(hit->tb, hit->qb);
			uint32_t x2 = (uint32_t)ORA_MIN(((int)reads[hit->pb1].len) - hit->te, ((int)reads[hit->pb2].len) - hit->qe);
			if(x1 + x2 <= C->P.max_unalign_in_dovetail){ C->rdcovs[hit->pb1]++; C->rdcovs[hit->pb2]++; }
			fprintf(out, "%s\t%c\t%d\t%d\t%d", reads[hit->pb1].name, '+', reads[hit->pb1].len, hit->tb, hit->te);
			fprintf(out, "\t%s\t%c\t%d\t%d\t%d", reads[hit->pb2].name, "+-"[hit->dir2], reads[hit->pb2].len, hit->qb, hit->qe);
			fprintf(out, "\t%d\t%0.3f\t%d\t%d\t%d\t%d", hit->score, 1.0 * hit->mat / hit->aln, hit->mat, hit->mis, hit->ins, hit->del);
			if(hit->cigar){ fprintf(out, "\t%s\n", hit->cigar); free(hit->cigar); hit->cigar = NULL; }
			else fprintf(out, "\t0M\n");
		}
	}
	K->hits.n = 0;
	return ret;
}

static uint64_t ora_commit(ora_ctx_t *C, ora_worker_t *K, FILE *out){
	uint64_t ret = 0;
	if(K->rd_id != 0xFFFFFFFFu) ret = ora_flush_hits(C, K, out);
	for(size_t i = 0; i < K->masks.n; i++) C->masked[K->masks.a[i]] = 1;
	K->masks.n = 0;
	for(size_t i = 0; i < K->closed.n; i++){
		if(!ora_u64set_has(&C->closed, K->closed.a[i])){
			uint64_t v = K->closed.a[i];
			uint32_t a = (uint32_t)(v >> 33), b = (uint32_t)((v & 0xFFFFFFFFu) >> 1);
			C->pair_bp += (uint64_t)C->st.reads.a[a].len + C->st.reads.a[b].len; C->n_pairs++;
		}
		ora_u64set_put(&C->closed, K->closed.a[i]);
	}
	K->closed.n = 0;
	return ret;
}

/* wtzmo.c:1251-1357 at one worker */
static uint64_t ora_overlap_all(ora_ctx_t *C, uint32_t n_job, uint32_t i_job, FILE *out){
	ora_worker_t *K = (ora_worker_t*)calloc(1, sizeof(ora_worker_t));
	uint64_t ret = 0;
	uint32_t n_rd = C->st.n_rd, n_idx = C->P.n_idx, pbbeg = 0, pbend = 0, beg, end;
	ora_kparams_t kp; kp.ksize = C->P.ksize; kp.hk = C->P.hk; kp.ksave = C->P.ksave; kp.kovl = C->P.kovl; kp.ncand = C->P.ncand; kp.max_kmer_freq = C->P.max_kmer_freq;
	K->rd_id = 0xFFFFFFFFu;
	if(n_idx > 1) C->rdhits = (vec_u64*)calloc((size_t)n_rd + C->n_qr, sizeof(vec_u64));
	for(uint32_t i_idx = 0; i_idx < n_idx; i_idx++){
		pbbeg = pbend;
		pbend = pbbeg + (n_rd + n_idx - 1) / n_idx;
		ora_kindex_build(&C->ix, &C->st, pbbeg, pbend, &kp);
		C->P.max_kmer_freq = kp.max_kmer_freq;
		if(C->n_qr){ uint64_t tot = 0; for(uint32_t i = 0; i < C->n_qr; i++) tot += C->st.reads.a[n_rd + i].len; C->ix.avg_rdlen = (uint32_t)(tot / C->n_qr); }
		if(i_idx + 1 >= n_idx) break;
		for(uint32_t j = 0; j < n_rd; j++){
			if((j % n_job) != i_job) continue;
			if(C->masked[j]) continue;
			ora_worker_run(C, K, j, K->bcov, 1);   /* just_query: bcov is whatever the worker last held (wtzmo.c:1293-1295) */
		}
	}
	if(C->n_qr == 0){ beg = 0; end = n_rd; } else { beg = n_rd; end = n_rd + C->n_qr; }
	for(uint32_t j = beg; j < end; j++){
		if((j % n_job) != i_job) continue;
		if(C->masked[j]) continue;
		ret += ora_commit(C, K, out);
		ora_worker_run(C, K, j, C->rdcovs[j], 0);
	}
	ret += ora_commit(C, K, out);
	return ret;
}

#endif
