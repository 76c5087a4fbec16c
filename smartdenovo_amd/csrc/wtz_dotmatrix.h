/*
 * wtz_dotmatrix.h — per-pair task body of the SW-free "dmo" engine (A7d).
 *
 *   wtz_denoise        hzm_aln.h:721-889    denoising_hzmps
 *   wtz_merge_blocks   hzm_aln.h:933-1054   fast_merge_wtseedv
 *   wtz_chain_blocks   hzm_aln.h:1056-1132  chaining_overhang_wtseedv
 *   wtz_dot_matrix_align hzm_aln.h:1134-1181
 *
 * The load-bearing quirks of the reference are kept (SURVEY §8a trap 3): the band loop compares a
 * diagonal index with the match count, never includes the last diagonal of a band, assumes that
 * same-strand matches of a diagonal are contiguous, and block merging treats every block as its own
 * diagonal.  Float arithmetic is written without contraction (the library is built -ffp-contract=off).
 */
#ifndef WTZ_DOTMATRIX_H
#define WTZ_DOTMATRIX_H

#include "wtz_window.h"

typedef struct { int32_t offset; uint32_t off, cnt; } wtz_diag_t;
typedef struct { int32_t score, qb, qe, tb, te, dir; } wtz_dm_result_t;

#define WTZ_SEED_OFF_MAX 0x7FFFFFFF

struct wtz_gt_zdiag { WTZ_HDM bool operator()(const wtz_zhit_t &a, const wtz_zhit_t &b) const {
	int64_t ka = ((((int64_t)ZH_OFF1(a)) - ((int64_t)ZH_OFF2(a))) << 32) | (int64_t)ZH_OFF1(a);
	int64_t kb = ((((int64_t)ZH_OFF1(b)) - ((int64_t)ZH_OFF2(b))) << 32) | (int64_t)ZH_OFF1(b); return ka > kb; } };
struct wtz_gt_idx_off1 { const wtz_zhit_t *rs; WTZ_HDM bool operator()(uint32_t a, uint32_t b) const { return ZH_OFF1(rs[a]) > ZH_OFF1(rs[b]); } };
struct wtz_gt_zgid { WTZ_HDM bool operator()(const wtz_zhit_t &a, const wtz_zhit_t &b) const {
	return a.gid > b.gid ? true : (a.gid < b.gid ? false : ZH_OFF1(a) > ZH_OFF1(b)); } };
struct wtz_gt_wdiag { WTZ_HDM bool operator()(const wtz_win_t &a, const wtz_win_t &b) const {
	int64_t ka = (((int64_t)(a.beg[0] - a.beg[1])) << 32) | (int64_t)a.beg[0];
	int64_t kb = (((int64_t)(b.beg[0] - b.beg[1])) << 32) | (int64_t)b.beg[0]; return ka > kb; } };
struct wtz_gt_widx_beg0 { const wtz_win_t *rs; WTZ_HDM bool operator()(uint32_t a, uint32_t b) const { return rs[a].beg[0] > rs[b].beg[0]; } };
struct wtz_gt_wgrp { WTZ_HDM bool operator()(const wtz_win_t &a, const wtz_win_t &b) const {
	return a.pb2 > b.pb2 ? true : (a.pb2 < b.pb2 ? false : a.beg[0] > b.beg[0]); } };
struct wtz_gt_wclosed { WTZ_HDM bool operator()(const wtz_win_t &a, const wtz_win_t &b) const { return a.closed > b.closed; } };
struct wtz_gt_wbeg0 { WTZ_HDM bool operator()(const wtz_win_t &a, const wtz_win_t &b) const { return a.beg[0] > b.beg[0]; } };

WTZ_HD void wtz_tidy_groups(uint32_t *g, uint32_t n){        /* hzm_aln.h:836-846 */
	for(uint32_t i = 1; i < n; i++){
		if(g[i] < i) continue;
		for(uint32_t j = i + 1; j < n; j++){
			if(g[j] != i) continue;
			for(uint32_t k = j + 1; k < n; k++) if(g[k] == j) g[k] = i;
		}
	}
}

/* band iteration over distinct diagonals: 0 finished, 1 skip (doff advanced), 2 process [doff, doff+dcnt) */
WTZ_HD int wtz_band_next(const wtz_diag_t *diags, uint32_t ndiag, uint32_t limit, uint32_t &doff, uint32_t &dcnt, int32_t &lst_offset, int32_t &end_offset, int32_t yvar){
	if(!(doff < limit)) return 0;
	lst_offset = diags[doff].offset;
	dcnt = 0;
	for(;;){
		if(diags[dcnt + doff].offset > lst_offset + yvar) break;
		if(dcnt + doff + 1 >= ndiag) break;
		dcnt++;
	}
	if(dcnt == 0) return 0;
	if(diags[doff + dcnt].offset == end_offset){ doff += dcnt; return 1; }
	end_offset = diags[doff + dcnt].offset;
	return 2;
}
WTZ_HD void wtz_band_advance(const wtz_diag_t *diags, uint32_t &doff, uint32_t dcnt, int32_t lst_offset, int32_t yvar){
	uint32_t i;
	for(i = doff; i < doff + dcnt; i++) if(diags[i].offset > lst_offset + yvar / 2) break;
	doff = i;
}

typedef struct { wtz_vec<wtz_zhit_t> dst; wtz_vec<wtz_win_t> regs[2]; wtz_vec<wtz_diag_t> diags; wtz_vec<uint32_t> block, grps; } wtz_dmscratch_t;

WTZ_HD void wtz_denoise(wtz_zhit_t *rs, uint32_t n_rs, wtz_dmscratch_t &S, int32_t xvar, int32_t yvar, int32_t min_linear_len, bool presorted){
	uint32_t i, j, k, doff, dcnt = 0, gid;
	int32_t len, lst, lst_offset = 0, end_offset;
	if(!presorted) wtz_sort_exact(rs, (size_t)n_rs, wtz_gt_zdiag());      /* hzm_aln.h:728; done by the wavefront when tie-free */
	S.diags.reserve(2); if(S.diags.a){ S.diags.a[0].offset = 0; S.diags.a[0].off = 0; S.diags.a[0].cnt = 0; }
	for(uint32_t dir = 0; dir < 2; dir++){
		S.diags.n = 0; S.dst.n = 0; S.regs[dir].n = 0;
		bool have = false;
		for(i = 0; i < n_rs; i++){
			if(ZH_STRAND(rs[i]) ^ dir) continue;
			int32_t dg = (int32_t)ZH_OFF1(rs[i]) - (int32_t)ZH_OFF2(rs[i]);
			if(have && S.diags.a[S.diags.n - 1].offset == dg) S.diags.a[S.diags.n - 1].cnt++;
			else { wtz_diag_t d; d.offset = dg; d.off = i; d.cnt = 1; if(!S.diags.push(d)) return; have = true; }
		}
		doff = 0; end_offset = -0x7FFFFFFF;
		S.grps.n = 0; S.grps.push(0);
		for(;;){
			int st = wtz_band_next(S.diags.a, S.diags.n, n_rs, doff, dcnt, lst_offset, end_offset, yvar);
			if(st == 0) break;
			if(st == 1) continue;
			S.block.n = 0;
			for(i = 0; i < dcnt; i++){
				const wtz_diag_t dg = S.diags.a[i + doff];
				for(j = 0; j < dg.cnt; j++){
					if(ZH_STRAND(rs[dg.off + j]) ^ dir) continue;
					if(!S.block.push(dg.off + j)) return;
				}
			}
			wtz_gt_idx_off1 g1; g1.rs = rs;
			wtz_sort_exact(S.block.a, (size_t)S.block.n, g1);
			int32_t p0_off1 = 0, p0_len1 = 0, p_off1, p_len1;
			if(S.block.n){ p0_off1 = (int32_t)ZH_OFF1(rs[S.block.a[0]]); p0_len1 = (int32_t)ZH_LEN1(rs[S.block.a[0]]); len = p0_len1; } else len = 0;
			j = 0;
			for(i = 1; i <= S.block.n; i++){
				if(i == S.block.n){ p_off1 = WTZ_SEED_OFF_MAX; p_len1 = 0; }
				else { p_off1 = (int32_t)ZH_OFF1(rs[S.block.a[i]]); p_len1 = (int32_t)ZH_LEN1(rs[S.block.a[i]]); }
				if(p_off1 <= p0_off1 + p0_len1){
					len += (p_off1 + p_len1) - (p0_off1 + p0_len1);
				} else if(p_off1 <= p0_off1 + p0_len1 + xvar){
					len += (p_off1 + p_len1) - (p0_off1 + p0_len1);
				} else {
					if(len >= min_linear_len){
						gid = 0;
						for(k = j; k < i; k++){
							uint32_t g = rs[S.block.a[k]].gid;
							if(g){ if(gid == 0) gid = S.grps.a[g]; else if(gid > S.grps.a[g]) gid = S.grps.a[g]; }
						}
						if(gid == 0){ gid = S.grps.n; if(!S.grps.push(gid)) return; }
						else { for(k = j; k < i; k++){ uint32_t g = rs[S.block.a[k]].gid; if(g) S.grps.a[g] = gid; } }
						for(; j < i; j++) rs[S.block.a[j]].gid = gid;
					}
					j = i;
					len = p0_len1;
				}
				p0_off1 = p_off1; p0_len1 = p_len1;
			}
			wtz_band_advance(S.diags.a, doff, dcnt, lst_offset, yvar);
		}
		wtz_tidy_groups(S.grps.a, S.grps.n);
		for(i = 0; i < n_rs; i++){
			if(ZH_STRAND(rs[i]) ^ dir) continue;
			if(rs[i].gid == 0) continue;
			rs[i].gid = S.grps.a[rs[i].gid];
			if(!S.dst.push(rs[i])) return;
		}
		wtz_sort_exact(S.dst.a, (size_t)S.dst.n, wtz_gt_zgid());
		j = 0;
		for(i = 1; i <= S.dst.n; i++){
			if(i < S.dst.n && S.dst.a[i].gid == S.dst.a[j].gid) continue;
			wtz_win_t seed;
			seed.pb2 = 0; seed.closed = 0; seed.dir = (uint8_t)dir; seed.pad = 0;
			seed.anchors[0] = j; seed.anchors[1] = i;
			seed.beg[0] = seed.beg[1] = 0x7FFFFFFF; seed.end[0] = seed.end[1] = 0; seed.ovl = 0;
			lst = 0;
			for(k = j; k < i; k++){
				const wtz_zhit_t p = S.dst.a[k];
				const int32_t o1 = (int32_t)ZH_OFF1(p), l1 = (int32_t)ZH_LEN1(p), o2 = (int32_t)ZH_OFF2(p), l2 = (int32_t)ZH_LEN2(p);
				if(o1 < seed.beg[0]) seed.beg[0] = o1;
				if(o1 + l1 > seed.end[0]) seed.end[0] = o1 + l1;
				if(o2 < seed.beg[1]) seed.beg[1] = o2;
				if(o2 + l2 > seed.end[1]) seed.end[1] = o2 + l2;
				seed.ovl = WTZ_OVL29(seed.ovl + (uint32_t)((o1 > lst) ? l1 : o1 + l1 - lst));
				lst = o1 + l1;
			}
			if(!(seed.end[0] - seed.beg[0] < min_linear_len)){ if(!S.regs[dir].push(seed)) return; }
			j = i;
		}
	}
}

WTZ_HD void wtz_merge_blocks(wtz_vec<wtz_win_t> &rv, wtz_dmscratch_t &S, int32_t xvar, int32_t yvar){
	wtz_win_t *regs = rv.a; const uint32_t n = rv.n;
	uint32_t i, j, k, doff, dcnt = 0, gid;
	int32_t lst_offset = 0, end_offset;
	wtz_sort_exact(regs, (size_t)n, wtz_gt_wdiag());
	S.diags.n = 0;
	if(!S.diags.reserve(n + 2)) return;
	for(i = 0; i < n; i++){ wtz_diag_t d; d.offset = regs[i].beg[0] - regs[i].beg[1]; d.off = i; d.cnt = 1; S.diags.a[S.diags.n++] = d; }
	doff = 0; end_offset = -0x7FFFFFFF;
	S.grps.n = 0; S.grps.push(0);
	for(;;){
		int st = wtz_band_next(S.diags.a, S.diags.n, n, doff, dcnt, lst_offset, end_offset, yvar);
		if(st == 0) break;
		if(st == 1) continue;
		S.block.n = 0;
		for(i = 0; i < dcnt; i++){ const wtz_diag_t dg = S.diags.a[i + doff]; for(j = 0; j < dg.cnt; j++) if(!S.block.push(dg.off + j)) return; }
		wtz_gt_widx_beg0 gb; gb.rs = regs;
		wtz_sort_exact(S.block.a, (size_t)S.block.n, gb);
		int32_t s0_end0 = S.block.n ? regs[S.block.a[0]].end[0] : 0;
		j = 0;
		for(i = 1; i <= S.block.n; i++){
			const int32_t s_beg0 = (i == S.block.n) ? WTZ_SEED_OFF_MAX : regs[S.block.a[i]].beg[0];
			const int32_t s_end0 = (i == S.block.n) ? 0 : regs[S.block.a[i]].end[0];
			if(s_beg0 <= s0_end0 + xvar){
			} else {
				gid = 0;
				for(k = j; k < i; k++){
					uint32_t g = regs[S.block.a[k]].pb2;
					if(g){ if(gid == 0) gid = S.grps.a[g]; else S.grps.a[g] = gid; }
				}
				if(gid == 0){ gid = S.grps.n; if(!S.grps.push(gid)) return; }
				for(; j < i; j++) regs[S.block.a[j]].pb2 = gid;
				j = i;
				s0_end0 = s_end0;
			}
		}
		wtz_band_advance(S.diags.a, doff, dcnt, lst_offset, yvar);
	}
	wtz_tidy_groups(S.grps.a, S.grps.n);
	for(i = 0; i < n; i++){ if(regs[i].pb2 == 0) continue; regs[i].pb2 = S.grps.a[regs[i].pb2]; }
	wtz_sort_exact(regs, (size_t)n, wtz_gt_wgrp());
	for(j = 0; j < n; j++) if(regs[j].pb2) break;
	for(i = j + 1; i <= n; i++){
		if(i < n && regs[i].pb2 == regs[j].pb2) continue;
		wtz_win_t *s0 = &regs[j];
		for(k = j + 1; k < i; k++){
			wtz_win_t *s = &regs[k];
			s->closed = 1;
			if(s->beg[0] < s0->beg[0]) s0->beg[0] = s->beg[0];
			if(s->end[0] > s0->end[0]) s0->end[0] = s->end[0];
			if(s->beg[1] < s0->beg[1]) s0->beg[1] = s->beg[1];
			if(s->end[1] > s0->end[1]) s0->end[1] = s->end[1];
			s0->ovl = WTZ_OVL29(s0->ovl + s->ovl);
		}
		j = i;
	}
	wtz_sort_exact(regs, (size_t)n, wtz_gt_wclosed());
	for(i = 0; i < n; i++) if(regs[i].closed) break;
	rv.n = i;
}

WTZ_HD int32_t wtz_w30(int32_t v){ return (int32_t)((uint32_t)v << 2) >> 2; }     /* node_t.weight:30, hzm_aln.h:1057 */

WTZ_HD int32_t wtz_chain_blocks(int32_t pblen1, int32_t pblen2, wtz_vec<wtz_win_t> &rv, wtz_pool_t *pool, int32_t tail_margin, int32_t max_overhang, float band_penalty, float gap_penalty, int32_t *bad){
	wtz_win_t *regs = rv.a; const uint32_t n = rv.n; uint32_t i, j;
	int32_t mw, bt, band, gap, weight, W, score;
	wtz_sort_exact(regs, (size_t)n, wtz_gt_wbeg0());
	int32_t *mem = (int32_t*)wtz_pool_alloc(pool, (size_t)(4 * n + 4) * 4);
	if(mem == NULL){ *bad = 1; return 0; }
	int32_t *nw = mem, *nbt = mem + n, *nhead = mem + 2 * n, *ntail = mem + 3 * n;
	for(i = 0; i < n; i++){
		nbt[i] = -1; nw[i] = 0; nhead[i] = 0; ntail[i] = 0;
		if(regs[i].beg[0] <= tail_margin || regs[i].beg[1] <= tail_margin) nhead[i] = 1;
		if(regs[i].end[0] + tail_margin > pblen1 || regs[i].end[1] + tail_margin > pblen2) ntail[i] = 1;
	}
	mw = -1000000; bt = -1;
	for(i = 0; i < n; i++){
		wtz_win_t *r1 = &regs[i];
		r1->closed = 1;
		nw[i] = wtz_w30(nw[i] + (int32_t)r1->ovl);
		weight = nw[i] * ((nhead[i] + 3) * (ntail[i] + 3)) / 16;
		if(weight > mw){ mw = weight; bt = (int32_t)i; }
		W = (int32_t)((float)nw[i] / gap_penalty);
		for(j = i + 1; j < n; j++){
			const wtz_win_t *r2 = &regs[j];
			if(r2->beg[0] + max_overhang < r1->end[0]) continue;
			if(r2->beg[1] + max_overhang < r1->end[1]) continue;
			if(r2->beg[0] - r1->end[0] > W) break;
			band = WTZ_ABSDIFF(r2->beg[0] - r1->end[0], r2->beg[1] - r1->end[1]);
			gap  = WTZ_MAX(r2->beg[0] - r1->end[0], r2->beg[1] - r1->end[1]);
			if(gap < 0) gap = -gap;
			float fa = (float)band * band_penalty, fb = (float)gap * gap_penalty;
			score = (int32_t)(fa + fb);
			score = nw[i] - score;
			if(nw[j] <= score){ nw[j] = wtz_w30(score); nbt[j] = (int32_t)i; nhead[j] = nhead[i]; }
		}
	}
	mw = 0;
	while(bt >= 0){ regs[bt].closed = 0; mw += (int32_t)regs[bt].ovl; bt = nbt[bt]; }
	return mw;
}

WTZ_HD wtz_dm_result_t wtz_dot_matrix_align(wtz_vec<wtz_zhit_t> &cache, wtz_pool_t *pool, int32_t pblen1, int32_t pblen2, const wtz_params_t *P, int32_t *bad, bool presorted, uint64_t *tick_denoise){
	wtz_dm_result_t ret; int32_t weight[2]; uint32_t d;
	wtz_dmscratch_t S;
	S.dst.init(pool, cache.n / 2 + 16); S.regs[0].init(pool, 16); S.regs[1].init(pool, 16);
	S.diags.init(pool, 64); S.block.init(pool, 64); S.grps.init(pool, 16);
	/* dst of strand 0 is consumed before strand 1 reuses it: regs only keep bounds */
	wtz_denoise(cache.a, cache.n, S, P->xvar, P->yvar, P->min_block_len, presorted);
#if defined(__HIP_DEVICE_COMPILE__)
	*tick_denoise = (uint64_t)clock64();
#else
	*tick_denoise = 0;
#endif
	wtz_merge_blocks(S.regs[0], S, P->xvar, 2 * P->yvar);
	wtz_merge_blocks(S.regs[1], S, P->xvar, 2 * P->yvar);
	weight[0] = wtz_chain_blocks(pblen1, pblen2, S.regs[0], pool, P->xvar, P->max_overhang, P->deviation_penalty, P->gap_penalty, bad);
	weight[1] = wtz_chain_blocks(pblen1, pblen2, S.regs[1], pool, P->xvar, P->max_overhang, P->deviation_penalty, P->gap_penalty, bad);
	if(S.dst.bad || S.regs[0].bad || S.regs[1].bad || S.diags.bad || S.block.bad || S.grps.bad) *bad = 1;
	d = (weight[0] < weight[1]);
	ret.score = weight[d];
	ret.qb = ret.tb = 0x7FFFFFFF; ret.qe = ret.te = 0;
	for(uint32_t i = 0; i < S.regs[d].n; i++){
		const wtz_win_t &s = S.regs[d].a[i];
		if(s.closed == 0){
			if(ret.qb > s.beg[1]) ret.qb = s.beg[1];
			if(ret.tb > s.beg[0]) ret.tb = s.beg[0];
			if(ret.qe < s.end[1]) ret.qe = s.end[1];
			if(ret.te < s.end[0]) ret.te = s.end[0];
		}
	}
	ret.dir = (int32_t)d;
	return ret;
}

#endif
