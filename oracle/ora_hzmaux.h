/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see ora_util.h): CPU restatement of align_hzmaux, the pair routine of wtgbo
 * (SURVEY §8f1).  Parity is PINNED: tests/test_oracle_functions.py runs it beside the reference's own align_hzmaux
 * (oracle/_ref/libref_shim.so, compiled from /root/reference/hzm_aln.h) on seeded read pairs, and oracle/wtgbo_oracle — the
 * whole program over this routine — equals the goldens of the real `wtgbo -t 1`.
 *
 *   ora_hzmaux_index   <- reset_hzmaux + add_tseq_hzmaux + ready_hzmaux   hzm_aln.h:1664-1682, 70-115
 *   ora_align_hzmaux   <- align_hzmaux                                     hzm_aln.h:1684-1775
 *
 * Only the whole-read form (beg = end = 0, wtgbo.c:51) is restated: query_single_read_seeds_by_region (hzm_aln.h:117-171) with
 * qb = 0, qe = rdlen, tb = 0, te = tlen emits exactly what query_single_read_seeds emits, and filter_by_region_hzmps
 * (hzm_aln.h:1188-1197) then only drops the matches of the other strand.
 */
#ifndef ORA_HZMAUX_H
#define ORA_HZMAUX_H

#include "ora_overlap.h"

typedef struct {
	uint32_t zsize, hz, zwin, zstep, zovl, zmax, zvar;
	int w, W, ew, rw, M, X, I, D, E, T;
} ora_auxparams_t;

static void ora_auxparams_wtgbo(ora_auxparams_t *p){       /* wtgbo.c:385-411, 470-486 */
	p->zsize = 10; p->hz = 1; p->zwin = 800; p->zstep = 0; p->zovl = 200; p->zmax = 100; p->zvar = 2;
	p->w = 50; p->W = 3200; p->ew = 800; p->rw = 50; p->M = 2; p->X = -5; p->I = -3; p->D = -3; p->E = -1; p->T = -50;
}

typedef struct {
	vec_u8 tseq; ora_ztable_t zt;
	vec_u8 kcnts; vec_zhit all, rs, anchors; vec_win windows; ora_winscratch_t wsc; vec_i32 chainmem;
	vec_u32 cigar_cache, cigars, tmp_cigar; vec_reg regs; ora_swmem_t swmem; ora_refmem_t refmem;
	ora_aln_t hit;
} ora_hzmaux_t;

static void ora_hzmaux_index(ora_hzmaux_t *A, const ora_auxparams_t *P, const uint8_t *tseq, uint32_t tlen){
	vec_u8_reserve(&A->tseq, (size_t)tlen + 8); memcpy(A->tseq.a, tseq, tlen); A->tseq.n = tlen;
	ora_ztable_build(&A->zt, A->tseq.a, tlen, P->zsize, (int)P->hz, P->zmax);
}

/* 1 = hit (A->hit, A->cigars), 0 = none */
static int ora_align_hzmaux(ora_hzmaux_t *A, const ora_auxparams_t *P, const uint8_t *rdseq, int rdlen, int refine_align, float min_sm){
	const int tlen = (int)A->tseq.n;
	ora_zmatch(&A->zt, rdseq, (uint32_t)rdlen, P->zsize, (int)P->hz, P->zmax, P->zvar, &A->kcnts, &A->all);
	/* process_hzmps on ALL matches (hzm_aln.h:1694), then the strand / region filter (1695) */
	vec_zhit_reserve(&A->all, A->all.n + 1); memset(&A->all.a[A->all.n], 0, sizeof(ora_zhit_t));
	ora_sort_zhit_off12(A->all.a, A->all.n, NULL);
	A->rs.n = 0;
	for(size_t i = 0; i < A->all.n; i++){
		const ora_zhit_t *p = &A->all.a[i];
		if(p->dir1 ^ p->dir2) continue;
		if((p->off1 + p->len1 > (uint32_t)tlen) || (p->off2 + p->len2 > (uint32_t)rdlen)) continue;
		vec_zhit_push(&A->rs, *p);
	}
	vec_zhit_reserve(&A->rs, A->rs.n + 1); memset(&A->rs.a[A->rs.n], 0, sizeof(ora_zhit_t));
	A->windows.n = 0; A->anchors.n = 0;
	if(ora_merge_windows(A->rs.a, (uint32_t)A->rs.n, 0, &A->windows, &A->anchors, &A->wsc, P->zsize, P->zwin, P->zstep, P->zovl) == 0) return 0;
	if(ora_chain_windows(A->windows.a, 0, (uint32_t)A->windows.n, P->W, &A->chainmem) < (int)P->zovl) return 0;
	A->regs.n = 0; A->cigar_cache.n = 0;
	for(size_t i = 0; i < A->windows.n; i++){
		const ora_win_t *seed = &A->windows.a[i];
		if(seed->closed) continue;
		vec_u32_push(&A->cigar_cache, (0u << 4) | 0xFu);
		ora_reg_t R;
		R.cigar_off = (uint32_t)A->cigar_cache.n;
		R.x = ora_align_window(A->tseq.a, rdseq, seed, A->anchors.a, &A->cigar_cache, &A->swmem, &A->tmp_cigar, P->w, P->M, P->X, P->I, P->D, P->E, P->T);
		R.cigar_len = (uint32_t)A->cigar_cache.n - R.cigar_off;
		if(R.x.aln * 2 < (int)P->zovl || R.x.mat < R.x.aln * min_sm) continue;
		vec_reg_push(&A->regs, R);
	}
	if(A->regs.n == 0) return 0;
	int esti[2] = {0, tlen};
	ora_aln_t x = ora_stitch_windows(tlen, rdlen, A->regs.a, A->regs.n, esti, A->tseq.a, rdseq, A->cigar_cache.a, &A->cigars, &A->swmem, &A->tmp_cigar,
		P->W, P->ew, P->w, P->M, P->X, P->I, P->D, P->E, P->T);
	int beg = x.qb - x.tb; if(beg < 0) beg = 0;
	int end = x.qe + tlen - x.te; if(end > rdlen) end = rdlen;
	const int ovl = end - beg;
	if(x.score < 0 || x.mat < x.aln * min_sm || x.mat < ovl * min_sm) return 0;
	if(refine_align){
		A->cigar_cache.n = 0; vec_u32_append(&A->cigar_cache, A->cigars.a, A->cigars.n);
		x = ora_refine_alignment(rdseq, x.qb, A->tseq.a, x.tb, P->rw, P->M, P->X, P->I, P->D, P->E, A->cigar_cache.a, A->cigar_cache.n, &A->refmem, &A->cigars);
	}
	A->hit = x;
	return 1;
}

#endif
