/*
 * wtz_dotmatrix.h — per-pair task body of the SW-free "dmo" engine (A7d).
 *
 *   wtz_denoise        hzm_aln.h:721-889    denoising_hzmps
 *   wtz_merge_blocks   hzm_aln.h:933-1054   fast_merge_wtseedv (wtz_blockband + wtz_blocklabels, one lane)
 *   wtz_chain_blocks_coop hzm_aln.h:1056-1132  chaining_overhang_wtseedv (the whole wavefront: one step per block)
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

/* one strand of denoising_hzmps (hzm_aln.h:731-889) over the diagonal-ordered matches */
WTZ_HD void wtz_denoise_dir(wtz_zhit_t *rs, uint32_t n_rs, uint32_t dir, wtz_dmscratch_t &S, int32_t xvar, int32_t yvar, int32_t min_linear_len){
	uint32_t i, j, k, doff, dcnt = 0, gid;
	int32_t len, lst, lst_offset = 0, end_offset;
	S.diags.reserve(2); if(S.diags.a && S.diags.n == 0){ S.diags.a[0].offset = 0; S.diags.a[0].off = 0; S.diags.a[0].cnt = 0; }
	{
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

/*
 * The same strand pass with the wavefront and LDS (same results).  denoising_hzmps is a chain of small order-sensitive
 * steps; run from HBM by one lane it costs ~10 dependent loads per match.  Here all lanes build a compact image of the
 * strand in LDS - per match (off1<<10 | len1), its index in rs, its group id; per distinct diagonal (offset, first match,
 * number of band members) - and lane 0 runs the band loop / group merging on that image only.  The grouped matches are
 * then ordered by (group, off1) with the wave-wide bitonic network (an equal key makes lane 0 redo the swap-exact sort
 * from the original order), gathered into LDS and folded into blocks.  Returns non-zero without having changed anything
 * when the strand does not fit (the caller retries with a larger slice, then the `big` form, then wtz_denoise_dir on lane 0).
 */
#ifndef WTZ_DM_BCAP_BIG_AT
#define WTZ_DM_BCAP_BIG_AT 49152u
#endif
#define WTZ_DM_BCAP(lds_bytes) ((lds_bytes) >= WTZ_DM_BCAP_BIG_AT ? 2048u : ((lds_bytes) >= 24576u ? 1024u : 512u))      /* members of one diagonal band the LDS list holds */
#define WTZ_DM_BCAP_MIN 512u
#define WTZ_DM_GCAP 255u      /* linear groups of one strand (one byte per match) */
struct wtz_gt_blk_off1 { const uint32_t *T; WTZ_HDM bool operator()(uint16_t a, uint16_t b) const { return (T[a] >> 10) > (T[b] >> 10); } };
struct wtz_gt_hi48 { WTZ_HDM bool operator()(uint64_t a, uint64_t b) const { return (a >> 16) > (b >> 16); } };

/* matches (nf) and distinct diagonals (nd) of both strands of the diagonal-ordered match list, in one pass of the wave:
 * a match opens a new diagonal iff the previous match OF ITS STRAND has another offset (hzm_aln.h:733-744) */
typedef struct { uint32_t nf[2], nd[2]; } wtz_dm_counts_t;
WTZ_HD wtz_dm_counts_t wtz_dm_counts(const wtz_zhit_t *rs, uint32_t n_rs){
	wtz_dm_counts_t c; c.nf[0] = c.nf[1] = c.nd[0] = c.nd[1] = 0;
	const uint32_t lane = WTZ_LANE;
	int32_t last_dg[2] = {0, 0}; uint32_t have[2] = {0, 0};
	for(uint32_t b0 = 0; b0 < n_rs; b0 += 4 * WTZ_NLANES){
		wtz_zhit_t hh[4];
		#pragma unroll
		for(int u = 0; u < 4; u++){ const uint32_t idx = b0 + u * WTZ_NLANES + lane; if(idx < n_rs) hh[u] = rs[idx]; else { hh[u].o1 = hh[u].o2 = hh[u].ll = hh[u].gid = 0; } }
		#pragma unroll
		for(int u = 0; u < 4; u++){
			const uint32_t idx = b0 + u * WTZ_NLANES + lane;
			const bool in = idx < n_rs;
			const uint32_t st = ZH_STRAND(hh[u]);
			const int32_t dg = (int32_t)ZH_OFF1(hh[u]) - (int32_t)ZH_OFF2(hh[u]);
			#pragma unroll
			for(uint32_t dir = 0; dir < 2; dir++){
				const bool keep = in && st == dir;
#if defined(__HIP_DEVICE_COMPILE__)
				const unsigned long long m = __ballot(keep);
				if(m == 0) continue;
				const unsigned long long below = m & ((1ull << lane) - 1ull);
				const int prevl = below ? 63 - __clzll((long long)below) : -1;
				const int32_t pdg = __shfl(dg, prevl < 0 ? 0 : prevl, 64);
				const bool head = keep && (prevl >= 0 ? (pdg != dg) : (!have[dir] || last_dg[dir] != dg));
				c.nf[dir] += (uint32_t)__popcll(m);
				c.nd[dir] += (uint32_t)__popcll(__ballot(head));
				last_dg[dir] = __shfl(dg, 63 - __clzll((long long)m), 64); have[dir] = 1;
#else
				if(keep){ c.nf[dir]++; if(!have[dir] || last_dg[dir] != dg) c.nd[dir]++; last_dg[dir] = dg; have[dir] = 1; }
#endif
			}
		}
	}
	return c;
}
/* Strand image of wtz_denoise_dir_coop (round-4 layout).  Per match 4 B (off1 << 10 | len1) + the group id (1 B; 2 B in the `big` form); per distinct
 * diagonal 2 B (first match) + 2 B (matches the band loop takes) + 4 B (offset).  The OFFSETS are only read while the band list is made, the group ids and
 * the band work arrays only afterwards - so the offsets overlay those two (ids are zeroed once the list stands): 8 B per match + 4 B per diagonal instead of
 * 5 + 8, i.e. at nd ~ nf a 24 KB slice holds strands of ~2 000 matches instead of ~1 500.  gw = bytes of a group id. */
typedef struct { uint32_t off_fo, off_mc, off_g, off_w, total; } wtz_dm_layout_t;
WTZ_HD wtz_dm_layout_t wtz_dm_layout(uint32_t nf, uint32_t nd, uint32_t gw, uint32_t wrk_bytes){
	wtz_dm_layout_t L;
	L.off_fo = 4u * (nf + 2u);
	L.off_mc = L.off_fo + 2u * (nd + 2u);
	L.off_g = (L.off_mc + 2u * (nd + 2u) + 3u) & ~3u;
	L.off_w = (L.off_g + gw * (nf + 4u) + 7u) & ~7u;
	const uint32_t dend = L.off_g + 4u * (nd + 2u);              /* end of the offsets, which start where the group ids start */
	L.total = L.off_w + wrk_bytes > dend ? L.off_w + wrk_bytes : dend;
	return L;
}
/* LDS bytes of the strand image + work arrays in the ordinary form (one-byte group ids) */
WTZ_HD uint32_t wtz_denoise_lds_need(uint32_t nf, uint32_t nd, uint32_t lds_bytes){
	(void)lds_bytes;
	return wtz_dm_layout(nf, nd, 1u, 8u * WTZ_DM_BCAP_MIN + 2u * WTZ_DM_GCAP + 40u).total;      /* with the smallest band list; a strand takes the largest that fits */
}

#define WTZ_DM_GCAP_BIG 8191u
/* Everything of wtz_denoise_dir_coop behind the layout decision, compiled twice: IMG_LDS = the strand image is in the wave's LDS slice (the usual case),
 * else in the pool.  One body with `img = fits ? lds : pool` left every access to the image a FLAT instruction - the address space of a selected pointer is
 * unknown - i.e. 39 G flat loads / stores per configs[2] dmo step, each through the texture-address path although nine strands in ten hit LDS (round-4 PMC
 * pass: 76 % of K_pair_dm's wave cycles waiting, the same 5.2 s with six or eight waves per CU).  With the pointer's origin a compile-time fact the LDS
 * form uses ds_read / ds_write. */
template<bool IMG_LDS>
WTZ_HD int wtz_denoise_dir_body(const wtz_zhit_t *rs, uint32_t n_rs, uint32_t dir, uint32_t nf, uint32_t nd, wtz_dmscratch_t &S, int32_t xvar, int32_t yvar, int32_t min_linear_len,
		uint8_t *lds, uint32_t lds_bytes, wtz_pool_t *pool, int32_t *bad, bool big, uint32_t bcap, uint32_t gcap, const wtz_dm_layout_t LY, uint8_t *img_pool, unsigned long long pd0, unsigned long long pdn){
	const uint32_t lane = WTZ_LANE;
	uint8_t *img = IMG_LDS ? lds : img_pool, *wrk = IMG_LDS ? lds + LY.off_w : lds;
	(void)lds_bytes; (void)pd0;
	uint64_t ra = 0;
	if(lane == 0) ra = (uint64_t)(uintptr_t)wtz_pool_alloc(pool, (size_t)(nf + 2u) * 2u);
	ra = wtz_coop_bcast64(ra);
	uint16_t *ridx = (uint16_t*)(uintptr_t)ra;            /* index in rs (pool) */
	if(ridx == NULL){ *bad = 1; if(lane == 0) S.regs[dir].n = 0; return 0; }
	uint32_t *T = (uint32_t*)img;
	uint8_t *gid8 = img + LY.off_g; uint16_t *gid16 = (uint16_t*)(img + LY.off_g);      /* group id of the match (valid once the band list is made: the offsets lie here until then) */
#define WTZ_GID(i) (big ? (uint32_t)gid16[i] : (uint32_t)gid8[i])
#define WTZ_GID_SET(i, v) do { if(big) gid16[i] = (uint16_t)(v); else gid8[i] = (uint8_t)(v); } while(0)
	int32_t *Doff = (int32_t*)(img + LY.off_g);           /* diagonal offset: over the group ids and (LDS image) the band work arrays, dead before either is first written */
	uint16_t *Dfo = (uint16_t*)(img + LY.off_fo);         /* first match (strand-compacted position) */
	uint16_t *Dmc = (uint16_t*)(img + LY.off_mc);         /* matches the band loop takes from it */
	uint32_t *bk = (uint32_t*)(((uintptr_t)wrk + 3u) & ~(uintptr_t)3u);      /* band keys; later run heads */
	uint16_t *blk = (uint16_t*)(bk + bcap);               /* band members, diagonal order */
	uint16_t *sblk = blk + bcap;                           /* band members, off1 order */
	uint16_t *grp = sblk + bcap;
	{   /* one pass: matches of the strand compacted in order, diagonal heads (hzm_aln.h:733-744) ranked in order */
		uint32_t n = 0, h = 0; int32_t last_dg = 0; uint32_t have = 0;
		for(uint32_t b0 = 0; b0 < n_rs; b0 += 4 * WTZ_NLANES){
			wtz_zhit_t hh[4];
			#pragma unroll
			for(int u = 0; u < 4; u++){ const uint32_t idx = b0 + u * WTZ_NLANES + lane; if(idx < n_rs) hh[u] = rs[idx]; else { hh[u].o1 = hh[u].o2 = hh[u].ll = hh[u].gid = 0; } }
			#pragma unroll
			for(int u = 0; u < 4; u++){
				const uint32_t idx = b0 + u * WTZ_NLANES + lane;
				const bool keep = idx < n_rs && ZH_STRAND(hh[u]) == dir;
				const int32_t dg = (int32_t)ZH_OFF1(hh[u]) - (int32_t)ZH_OFF2(hh[u]);
				uint32_t tot; const uint32_t pos = wtz_coop_rank(keep, &tot);
				if(tot == 0) continue;
				bool head;
#if defined(__HIP_DEVICE_COMPILE__)
				{
					const unsigned long long m = __ballot(keep);
					const unsigned long long below = m & ((1ull << lane) - 1ull);
					const int prevl = below ? 63 - __clzll((long long)below) : -1;
					const int32_t pdg = __shfl(dg, prevl < 0 ? 0 : prevl, 64);
					head = keep && (prevl >= 0 ? (pdg != dg) : (!have || last_dg != dg));
					last_dg = __shfl(dg, 63 - __clzll((long long)m), 64); have = 1;
				}
#else
				head = keep && (!have || last_dg != dg);
				if(keep){ last_dg = dg; have = 1; }
#endif
				uint32_t htot; const uint32_t hpos = wtz_coop_rank(head, &htot);
				if(keep){ T[n + pos] = (ZH_OFF1(hh[u]) << 10) | (ZH_LEN1(hh[u]) & 0x3FFu); ridx[n + pos] = (uint16_t)idx; }
				if(head){ Doff[h + hpos] = dg; Dfo[h + hpos] = (uint16_t)(n + pos); }
				n += tot; h += htot;
			}
		}
		if(lane == 0){ Dfo[nd] = (uint16_t)nf; Doff[nd] = 0; if(nd == 0){ Doff[0] = 0; Dfo[0] = 0; Dmc[0] = 0; } }
	}
	WTZ_WAVE_SYNC();
	/* the reference walks cnt consecutive rs entries from the diagonal's first match and keeps those of the strand
	 * (hzm_aln.h:771-776): of the diagonal's matches, the ones whose rs index is below first + cnt */
	for(uint32_t h0 = 0; h0 < nd; h0 += WTZ_NLANES){
		const uint32_t h = h0 + lane;
		if(h < nd){
			const uint32_t fo = Dfo[h], cnt = (uint32_t)Dfo[h + 1] - fo;
			uint32_t m = cnt;
			if(cnt > 1u){
				const uint32_t base = ridx[fo];
				if((uint32_t)ridx[fo + cnt - 1u] - base >= cnt){ m = 0; while(m < cnt && (uint32_t)ridx[fo + m] - base < cnt) m++; }      /* other-strand matches in between */
			}
			Dmc[h] = (uint16_t)m;
		}
	}
	WTZ_WAVE_SYNC();
	WTZ_PROF_ADD(24, pdn); pdn = WTZ_PROF_T();
	/* ---- the bands.  Their sequence (hzm_aln.h:748-769, 832-834) depends on the diagonal offsets only, so lane 0 lists
	 * them first; whether a band can change anything - it must contain a linear run of >= min_linear_len - is then decided
	 * for all bands in parallel (small bands exactly, by replaying the sweep in registers; larger ones are simply kept), and
	 * only the productive bands go through the sequential group merging. ---- */
	uint64_t ba = 0;
	if(lane == 0) ba = (uint64_t)(uintptr_t)wtz_pool_alloc(pool, (size_t)(nd + 2u) * 8u);
	ba = wtz_coop_bcast64(ba);
	uint32_t *bands = (uint32_t*)(uintptr_t)ba;           /* doff<<16 | dcnt per band, then the productive subset */
	if(bands == NULL){ *bad = 1; if(lane == 0) S.regs[dir].n = 0; return 0; }
	uint32_t *prod = bands + (nd + 2u);
	uint32_t nbands = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	if constexpr(IMG_LDS){
		/* Round 5: the band sequence is a chain  start -> start + advance(start)  whose links depend on the diagonal offsets only: the length of the band that
		 * starts at diagonal d (first offset beyond Doff[d] + yvar, never the last diagonal: hzm_aln.h:757-761), its closing offset and the step to the next
		 * start (first offset beyond Doff[d] + yvar / 2) are computed for 64 starts at once - lane l searches for d = wb + l in the LDS image - and the chain
		 * is then followed through those registers: three v_readlane per band.  The round-4 form below paid two ballots, their scalar bit scans and a dozen
		 * dependent scalar instructions per band on the one scalar unit the waves of a CU share (800 cycles per band, 260 bands per strand: 30 % of the pass). */
		uint32_t doff = 0, wb = 0x80000000u; int32_t end_offset = -0x7FFFFFFF;
		uint32_t vd = 0, va = 0; int32_t vv = 0;
		while(nd != 0 && doff < n_rs){
			if(doff - wb >= 64u){
				wb = doff;
				const uint32_t d = wb + lane;
				vd = 0; va = 0; vv = 0;
				if(d < nd && yvar >= 0){
					const int32_t first = Doff[d];
					const uint32_t cap = nd - 1u - d;
					uint32_t lo = d + 1u, hi = nd;
					while(lo < hi){ const uint32_t mid = lo + (hi - lo) / 2u; if(Doff[mid] > first + yvar) hi = mid; else lo = mid + 1u; }
					const uint32_t dc = lo - d < cap ? lo - d : cap;
					lo = d; hi = d + dc;
					while(lo < hi){ const uint32_t mid = lo + (hi - lo) / 2u; if(Doff[mid] > first + yvar / 2) hi = mid; else lo = mid + 1u; }
					vd = dc; va = lo - d; vv = Doff[d + dc];
				}
			}
			const int r = (int)__builtin_amdgcn_readfirstlane((int)(doff - wb));
			const uint32_t dcnt = (uint32_t)__builtin_amdgcn_readlane((int)vd, r);
			if(dcnt == 0) break;
			const int32_t voff = __builtin_amdgcn_readlane(vv, r);
			if(voff == end_offset){ doff += dcnt; continue; }
			end_offset = voff;
			if(lane == 0) bands[nbands] = (doff << 16) | dcnt;
			nbands++;
			doff += (uint32_t)__builtin_amdgcn_readlane((int)va, r);
		}
	} else
	{
		/* Round 4: the same band sequence walked by the WHOLE wave.  Lane 0 alone paid ~12 dependent LDS reads per band (2 800 cycles; 308 bands per strand at
		 * configs[2]: a quarter of the denoise pass).  Here lane l holds Doff[doff + l]: the end of the band (first offset beyond lst + yvar, never the last
		 * diagonal: hzm_aln.h:757-761) and the next start (first offset beyond lst + yvar / 2) are two ballots over that window; bands of more than 64 diagonals
		 * take further windows.  Every quantity is wave-uniform; the quirks are those of wtz_band_next / wtz_band_advance. */
		uint32_t doff = 0, wb = 0; int32_t end_offset = -0x7FFFFFFF;
		int32_t w = lane < nd ? Doff[lane] : 0x7FFFFFFF;             /* register window: lane l holds Doff[wb + l]; a band start moves by a few diagonals, so one load serves ~10 bands
		                                                              * (a strand whose image lives in the pool pays an L2 round trip per load) */
		while(nd != 0 && doff < n_rs){                     /* hzm_aln.h:753 compares the diagonal index with rs->size */
			uint32_t r = doff - wb;
			if(r >= 32u){ wb = doff; r = 0; const uint32_t idx = wb + lane; w = idx < nd ? Doff[idx] : 0x7FFFFFFF; }
			const uint32_t cap = nd - 1u - doff;                  /* dcnt stops before the last diagonal (doff <= nd - 1 always) */
			const int32_t lst = __builtin_amdgcn_readlane(w, (int)__builtin_amdgcn_readfirstlane((int)r));
			uint32_t dcnt;
			{       /* the offsets ascend: lanes below r cannot exceed lst */
				const unsigned long long m = __ballot(w > lst + yvar);
				if(m){ dcnt = (uint32_t)__builtin_ctzll(m) - r; if(dcnt > cap) dcnt = cap; }
				else if(64u - r > cap) dcnt = cap;
				else {
					uint32_t c = 64u - r;
					for(;;){
						const uint32_t idx = doff + c + lane;
						const int32_t v = idx < nd ? Doff[idx] : 0x7FFFFFFF;
						const unsigned long long m2 = __ballot(v > lst + yvar);
						if(m2){ dcnt = c + (uint32_t)__builtin_ctzll(m2); if(dcnt > cap) dcnt = cap; break; }
						if(c + 64u > cap){ dcnt = cap; break; }
						c += 64u;
					}
				}
			}
			if(dcnt == 0) break;
			int32_t voff = r + dcnt < 64u ? __builtin_amdgcn_readlane(w, (int)__builtin_amdgcn_readfirstlane((int)(r + dcnt))) : Doff[doff + dcnt];      /* Doff[doff + dcnt] */
			voff = __builtin_amdgcn_readfirstlane(voff);
			if(voff == end_offset){ doff += dcnt; continue; }
			end_offset = voff;
			if(lane == 0) bands[nbands] = (doff << 16) | dcnt;
			nbands++;
			uint32_t adv = dcnt;                                       /* wtz_band_advance: first diagonal of the band beyond lst + yvar / 2 */
			{
				const unsigned long long m = __ballot(w > lst + yvar / 2);
				if(m){ const uint32_t f = (uint32_t)__builtin_ctzll(m) - r; if(f < dcnt) adv = f; }
				else if(64u - r < dcnt){
					uint32_t c = 64u - r;
					for(;;){
						const uint32_t idx = doff + c + lane;
						const int32_t v = idx < nd ? Doff[idx] : 0x7FFFFFFF;
						const unsigned long long m2 = __ballot(v > lst + yvar / 2);
						if(m2){ const uint32_t f = c + (uint32_t)__builtin_ctzll(m2); if(f < dcnt) adv = f; break; }
						c += 64u;
						if(c >= dcnt) break;
					}
				}
			}
			doff += adv;
		}
	}
#else
	if(lane == 0){
		uint32_t doff = 0, dcnt = 0; int32_t lst_offset = 0, end_offset = -0x7FFFFFFF;
		for(;;){
			if(!(doff < n_rs)) break;                    /* wtz_band_next over Doff[] */
			lst_offset = Doff[doff];
			dcnt = 0;
			for(;;){
				if(Doff[dcnt + doff] > lst_offset + yvar) break;
				if(dcnt + doff + 1 >= nd) break;
				dcnt++;
			}
			if(dcnt == 0) break;
			if(Doff[doff + dcnt] == end_offset){ doff += dcnt; continue; }
			end_offset = Doff[doff + dcnt];
			bands[nbands++] = (doff << 16) | dcnt;
			uint32_t a;                                   /* wtz_band_advance */
			for(a = doff; a < doff + dcnt; a++) if(Doff[a] > lst_offset + yvar / 2) break;
			doff = a;
		}
	}
#endif
	nbands = wtz_coop_bcast32(nbands);
	WTZ_WAVE_SYNC();
	for(uint32_t x = lane; x < nf + 2u; x += WTZ_NLANES) WTZ_GID_SET(x, 0);      /* the offsets are dead: their bytes become the group ids (and the band work arrays) */
	WTZ_WAVE_SYNC();
	WTZ_PROF_ADD(25, pdn); pdn = WTZ_PROF_T(); WTZ_PROF_CNT(30, nbands);
	uint32_t nprod = 0;
	for(uint32_t b0 = 0; b0 < nbands; b0 += WTZ_NLANES){
		const uint32_t b = b0 + lane;
		bool keep = false;
		if(b < nbands){
			const uint32_t doff = bands[b] >> 16, dcnt = bands[b] & 0xFFFFu;
			uint32_t nb = 0;
			for(uint32_t i = 0; i < dcnt && nb <= 8u; i++) nb += Dmc[doff + i];
			if(nb > 8u) keep = true;
			else if(nb){
				/* the <= 8 members into eight NAMED registers (every index below is a compile-time constant after unrolling: an insertion sort into v[q]
				 * with a run-time q put the array in scratch memory - this loop was 1.4 of the 10 Tcycles of the pass), ordered by a 19-exchange network;
				 * members that tie on off1 carry identical words, so comparing whole words orders by off1 */
				uint32_t v[8]; const uint32_t m = nb;
				{
					uint32_t ci = 0, cj = 0;
					#pragma unroll
					for(int s2 = 0; s2 < 8; s2++){
						v[s2] = 0xFFFFFFFFu;
						if((uint32_t)s2 < nb){ while(cj >= (uint32_t)Dmc[doff + ci]){ ci++; cj = 0; } v[s2] = T[(uint32_t)Dfo[doff + ci] + cj]; cj++; }
					}
				}
#define WTZ_CE(A, B) do { const uint32_t lo_ = v[A] < v[B] ? v[A] : v[B], hi_ = v[A] < v[B] ? v[B] : v[A]; v[A] = lo_; v[B] = hi_; } while(0)
				WTZ_CE(0, 1); WTZ_CE(2, 3); WTZ_CE(4, 5); WTZ_CE(6, 7); WTZ_CE(0, 2); WTZ_CE(1, 3); WTZ_CE(4, 6); WTZ_CE(5, 7); WTZ_CE(1, 2); WTZ_CE(5, 6);
				WTZ_CE(0, 4); WTZ_CE(1, 5); WTZ_CE(2, 6); WTZ_CE(3, 7); WTZ_CE(2, 4); WTZ_CE(3, 5); WTZ_CE(1, 2); WTZ_CE(3, 4); WTZ_CE(5, 6);
#undef WTZ_CE
				#pragma unroll
				for(int i = 0; i + 1 < 8; i++) if((uint32_t)(i + 1) < m && (v[i] >> 10) == (v[i + 1] >> 10)) keep = true;      /* equal off1: left to the wave path (conservative: it would be harmless here too) */
				if(!keep){
					int32_t p0o = (int32_t)(v[0] >> 10), p0l = (int32_t)(v[0] & 0x3FFu), len = p0l;
					#pragma unroll
					for(int i = 1; i <= 8; i++){
						if((uint32_t)i <= m){
							const bool end = (uint32_t)i == m;
							const int32_t po = end ? WTZ_SEED_OFF_MAX : (int32_t)(v[i < 8 ? i : 7] >> 10), pl = end ? 0 : (int32_t)(v[i < 8 ? i : 7] & 0x3FFu);
							if(po <= p0o + p0l || po <= p0o + p0l + xvar) len += (po + pl) - (p0o + p0l);
							else { if(len >= min_linear_len) keep = true; len = p0l; }
							p0o = po; p0l = pl;
						}
					}
				}
			}
		}
		uint32_t tot; const uint32_t pos = wtz_coop_rank(keep, &tot);
		if(keep) prod[nprod + pos] = bands[b];
		nprod += tot;
	}
	WTZ_WAVE_SYNC();
	/* ---- productive bands in order, each on the whole wave:
	 *   members (diagonal order) -> bk[] = off1<<11 | position in the band, bitonic in LDS -> members by off1; an equal off1
	 *   makes lane 0 run the swap-exact sort instead (hzm_aln.h:777);
	 *   run breaks are local (a member starts a run iff it lies more than xvar beyond the previous member's end), and the
	 *   length the reference accumulates for run [j,i) telescopes to len1(j-1) + end(i-1) - end(j) (len1(0) for the first
	 *   run: hzm_aln.h:780-828 resets `len` to the PREVIOUS match's length) - so all runs of the band are measured at once;
	 *   only the runs of >= min_linear_len matches' worth go through the sequential group merging, each as a wave-wide
	 *   min-reduction + scatter. ---- */
	WTZ_PROF_ADD(26, pdn); pdn = WTZ_PROF_T(); WTZ_PROF_CNT(31, nprod);
	uint32_t fail = 0, ngrp = 1;
	if(lane == 0) grp[0] = 0;
	WTZ_WAVE_SYNC();
	for(uint32_t pb = 0; pb < nprod && !fail; pb++){
		const uint32_t doff = prod[pb] >> 16, dcnt = prod[pb] & 0xFFFFu;
		unsigned long long pbn = WTZ_PROF_T(); (void)pbn;      /* profiler, inside a productive band: 46 members, 48 order by off1, 49 runs, 50 group merging; 54 members (count), 55 productive runs (count) */
		/* members in diagonal order */
		uint32_t nb = 0;
		for(uint32_t d0 = 0; d0 < dcnt; d0 += WTZ_NLANES){
			const uint32_t dd = d0 + lane;
			uint32_t fo = 0, mc = 0;
			if(dd < dcnt){ fo = Dfo[doff + dd]; mc = Dmc[doff + dd]; }
			uint32_t tot; const uint32_t ex = wtz_coop_excl_scan(mc, &tot);
			if(nb + tot <= bcap){ for(uint32_t j = 0; j < mc; j++) blk[nb + ex + j] = (uint16_t)(fo + j); }
			nb += tot;
		}
		if(nb > bcap){ fail = 2; WTZ_PROF_CNT(62, 1000000); break; }
		WTZ_WAVE_SYNC();
		WTZ_PROF_ADD(46, pbn); pbn = WTZ_PROF_T(); WTZ_PROF_CNT(54, nb);
		/* Members by off1 (hzm_aln.h:777-778).  The reference's sort is unstable, but members that tie on off1 are copies of ONE query z-mer (same off1 => same
		 * strand-filtered query position => same len1: their T words are identical), a run never breaks between them (the second starts at or before the
		 * first's end), and everything below reads T, or writes the SAME group id to every member of a run: no order among them can be observed.  So any
		 * sorted order is the reference's (rounds 2-3 replayed the swap sequence on lane 0 for every band with such a tie: 60 k cycles per productive band,
		 * a third of the whole denoise pass; -DWTZ_DM_EXACT_TIES keeps that form for cross-checking). */
#if defined(__HIP_DEVICE_COMPILE__)
		if(nb <= 64u){         /* the usual case (37 members on average at configs[2]): one key per lane, ordered in registers */
			const uint32_t kv = lane < nb ? (((T[blk[lane]] >> 10) << 11) | lane) : 0xFFFFFFFFu;
			const uint32_t sv = wtz_wave_sort32(kv);
#ifdef WTZ_DM_EXACT_TIES
			const uint32_t nx = (uint32_t)__shfl_down((int)sv, 1, 64);
			uint32_t any; (void)wtz_coop_rank(lane + 1 < nb && (sv >> 11) == (nx >> 11), &any);
			if(any){ if(lane == 0){ wtz_gt_blk_off1 g1; g1.T = T; wtz_sort_exact(blk, (size_t)nb, g1); } WTZ_WAVE_SYNC(); if(lane < nb) sblk[lane] = blk[lane]; }
			else
#endif
			if(lane < nb) sblk[lane] = blk[sv & 0x7FFu];
		} else
#endif
		{
			uint32_t np2 = 64; while(np2 < nb) np2 <<= 1;
			for(uint32_t i = lane; i < np2; i += WTZ_NLANES) bk[i] = i < nb ? (((T[blk[i]] >> 10) << 11) | i) : 0xFFFFFFFFu;
			WTZ_WAVE_SYNC();
			wtz_coop_sort_u32(bk, np2);
#ifdef WTZ_DM_EXACT_TIES
			bool tie = false;
			for(uint32_t i = lane; i + 1 < nb; i += WTZ_NLANES) if((bk[i] >> 11) == (bk[i + 1] >> 11)) tie = true;
			uint32_t any; (void)wtz_coop_rank(tie, &any);
			if(any){
				if(lane == 0){ wtz_gt_blk_off1 g1; g1.T = T; wtz_sort_exact(blk, (size_t)nb, g1); }
				WTZ_WAVE_SYNC();
				for(uint32_t i = lane; i < nb; i += WTZ_NLANES) sblk[i] = blk[i];
			} else
#endif
			for(uint32_t i = lane; i < nb; i += WTZ_NLANES) sblk[i] = blk[bk[i] & 0x7FFu];
		}
		WTZ_WAVE_SYNC();
		WTZ_PROF_ADD(48, pbn); pbn = WTZ_PROF_T();
		/* run heads: position 0 and every break; heads[] reuses the key words */
		uint16_t *heads = (uint16_t*)bk;
		uint32_t nrun = 0;
		for(uint32_t i0 = 0; i0 < nb; i0 += WTZ_NLANES){
			const uint32_t i = i0 + lane;
			bool head = false;
			if(i < nb){
				if(i == 0) head = true;
				else {
					const uint32_t tp = T[sblk[i - 1]], tc = T[sblk[i]];
					const int32_t pe = (int32_t)(tp >> 10) + (int32_t)(tp & 0x3FFu), o = (int32_t)(tc >> 10);
					head = !(o <= pe || o <= pe + xvar);
				}
			}
			uint32_t tot; const uint32_t pos = wtz_coop_rank(head, &tot);
			/* heads[] aliases bk[]: everything read from bk is already in sblk */
			if(head) heads[nrun + pos] = (uint16_t)i;
			nrun += tot;
		}
		WTZ_WAVE_SYNC();
		if(lane == 0) heads[nrun] = (uint16_t)nb;
		WTZ_WAVE_SYNC();
		/* productive runs, in order */
		uint16_t *pruns = heads + (nrun + 2u);
		uint32_t npr = 0;
		for(uint32_t r0 = 0; r0 < nrun; r0 += WTZ_NLANES){
			const uint32_t r = r0 + lane;
			bool keep = false;
			if(r < nrun){
				const uint32_t j = heads[r], i = heads[r + 1];
				const uint32_t tj = T[sblk[j]], tl = T[sblk[i - 1]];
				const int32_t l0 = (int32_t)(T[sblk[j ? j - 1 : 0]] & 0x3FFu);
				const int32_t len = l0 + ((int32_t)(tl >> 10) + (int32_t)(tl & 0x3FFu)) - ((int32_t)(tj >> 10) + (int32_t)(tj & 0x3FFu));
				keep = len >= min_linear_len;
			}
			uint32_t tot; const uint32_t pos = wtz_coop_rank(keep, &tot);
			if(keep) pruns[npr + pos] = (uint16_t)r;
			npr += tot;
		}
		WTZ_WAVE_SYNC();
		WTZ_PROF_ADD(49, pbn); pbn = WTZ_PROF_T(); WTZ_PROF_CNT(55, npr);
		for(uint32_t q = 0; q < npr && !fail; q++){
			const uint32_t r = pruns[q], j = heads[r], i = heads[r + 1];
			uint32_t gmin = 0xFFFFFFFFu;
			for(uint32_t k = j + lane; k < i; k += WTZ_NLANES){ const uint32_t g = WTZ_GID(sblk[k]); if(g){ const uint32_t v = grp[g]; gmin = v < gmin ? v : gmin; } }
			gmin = wtz_coop_min32(gmin);
			uint32_t g0;
			if(gmin == 0xFFFFFFFFu){
				if(ngrp >= gcap){ fail = 3; WTZ_PROF_CNT(63, 1000000); break; }
				g0 = ngrp; if(lane == 0) grp[ngrp] = (uint16_t)g0; ngrp++;
			} else {
				g0 = gmin;
				WTZ_WAVE_SYNC();
				for(uint32_t k = j + lane; k < i; k += WTZ_NLANES){ const uint32_t g = WTZ_GID(sblk[k]); if(g) grp[g] = (uint16_t)g0; }
			}
			WTZ_WAVE_SYNC();
			for(uint32_t k = j + lane; k < i; k += WTZ_NLANES) WTZ_GID_SET(sblk[k], g0);
			WTZ_WAVE_SYNC();
		}
		WTZ_PROF_ADD(50, pbn);
	}
	WTZ_PROF_ADD(27, pdn); pdn = WTZ_PROF_T();
	fail = wtz_coop_bcast32(fail);
	if(lane == 0 && !fail){      /* wtz_tidy_groups */
		for(uint32_t i = 1; i < ngrp; i++){
			if(grp[i] < i) continue;
			for(uint32_t j = i + 1; j < ngrp; j++){
				if(grp[j] != i) continue;
				for(uint32_t k = j + 1; k < ngrp; k++) if(grp[k] == j) grp[k] = (uint16_t)i;
			}
		}
	}
	fail = wtz_coop_bcast32(fail);
	if(fail) return (int)fail;
	WTZ_WAVE_SYNC();
	WTZ_PROF_ADD(51, pdn); WTZ_PROF_CNT(58, ngrp);      /* 51: tidy groups (lane 0), 52: count + allocation, 53: keys + order; 58 groups (count) */
	unsigned long long pgn = WTZ_PROF_T(); (void)pgn;
	/* ---- grouped matches, ordered by (group, off1) (hzm_aln.h:848-857) ---- */
	uint32_t n_dst = 0;
	for(uint32_t x0 = 0; x0 < nf; x0 += WTZ_NLANES){
		const uint32_t x = x0 + lane;
		const bool keep = x < nf && WTZ_GID(x) != 0;
		uint32_t tot; (void)wtz_coop_rank(keep, &tot); n_dst += tot;
	}
	if(lane == 0) S.regs[dir].n = 0;
	if(n_dst == 0) return 0;
	uint32_t np = 64; while(np < n_dst) np <<= 1;
	uint64_t ka = 0;
	if(lane == 0) ka = (uint64_t)(uintptr_t)wtz_pool_alloc(pool, (size_t)np * 8 + (size_t)(n_dst + 1) * sizeof(wtz_zhit_t));
	ka = wtz_coop_bcast64(ka);
	uint64_t *K = (uint64_t*)(uintptr_t)ka;
	if(K == NULL){ *bad = 1; return 0; }
	wtz_zhit_t *dstv = (wtz_zhit_t*)(K + np);
	WTZ_PROF_ADD(52, pgn); pgn = WTZ_PROF_T();
	/* keys (group, off1, position), then the order of hzm_aln.h:848-857.  Ties on (group, off1) are copies of one query z-mer again: what follows reads
	 * their off1 / len1 (equal) and takes minima / maxima over their candidate side, so their relative order is unobservable (see the band order above).
	 * The key words live in the pool; they are ordered through an LDS window - the band work arrays are free now (grp[] behind them is still read) -
	 * instead of 45 bitonic stages straight on HBM words (5.0 of the 22.8 Tcycles of the denoise pass at configs[2]). */
	{
		uint32_t n = 0;
		for(uint32_t x0 = 0; x0 < nf; x0 += WTZ_NLANES){
			const uint32_t x = x0 + lane;
			const bool keep = x < nf && WTZ_GID(x) != 0;
			uint32_t tot; const uint32_t pos = wtz_coop_rank(keep, &tot);
			if(keep) K[n + pos] = ((uint64_t)grp[WTZ_GID(x)] << 37) | ((uint64_t)(T[x] >> 10) << 16) | x;
			n += tot;
		}
		for(uint32_t i = n_dst + lane; i < np; i += WTZ_NLANES) K[i] = ~0ull;
		WTZ_WAVE_SYNC();
		uint64_t *win = (uint64_t*)(((uintptr_t)bk + 7u) & ~(uintptr_t)7u);
		const uint32_t avail = (8u * bcap - (uint32_t)((uintptr_t)win - (uintptr_t)bk)) / 8u;      /* bk + blk + sblk = 8 * bcap bytes */
		uint32_t ln = 128u; while(ln * 2u <= avail) ln <<= 1;
		wtz_coop_sort_u64_windowed(K, np, avail >= 128u ? win : NULL, ln);
#ifdef WTZ_DM_EXACT_TIES
		bool tie = false;
		for(uint32_t i = lane; i + 1 < n_dst; i += WTZ_NLANES) if((K[i] >> 16) == (K[i + 1] >> 16)) tie = true;
		uint32_t any; (void)wtz_coop_rank(tie, &any);
		if(any){
			n = 0;
			for(uint32_t x0 = 0; x0 < nf; x0 += WTZ_NLANES){
				const uint32_t x = x0 + lane;
				const bool keep = x < nf && WTZ_GID(x) != 0;
				uint32_t tot; const uint32_t pos = wtz_coop_rank(keep, &tot);
				if(keep) K[n + pos] = ((uint64_t)grp[WTZ_GID(x)] << 37) | ((uint64_t)(T[x] >> 10) << 16) | x;
				n += tot;
			}
			WTZ_WAVE_SYNC();
			if(lane == 0) wtz_sort_exact(K, (size_t)n_dst, wtz_gt_hi48());
			WTZ_WAVE_SYNC();
		}
#endif
	}
	WTZ_PROF_ADD(53, pgn);
	WTZ_PROF_ADD(28, pdn); pdn = WTZ_PROF_T(); WTZ_PROF_CNT(11, n_dst);
	/* the strand image is dead: gather the ordered matches (with their final group id) */
	for(uint32_t y = lane; y < n_dst; y += WTZ_NLANES){
		wtz_zhit_t h = rs[ridx[(uint32_t)(K[y] & 0xFFFFu)]];
		h.gid = (uint32_t)(K[y] >> 37);
		dstv[y] = h;
	}
	WTZ_WAVE_SYNC();
	const bool in_lds = (size_t)n_dst * sizeof(wtz_zhit_t) <= (size_t)lds_bytes;
	const wtz_zhit_t *dv = dstv;
	if(in_lds){
		wtz_zhit_t *L = (wtz_zhit_t*)lds;
		for(uint32_t y = lane; y < n_dst; y += WTZ_NLANES) L[y] = dstv[y];
		WTZ_WAVE_SYNC();
		dv = L;
	}
#if defined(__HIP_DEVICE_COMPILE__)
	{
		/* one seed per group (hzm_aln.h:858-889), the whole wave per group: bounds are minima / maxima over the members; the covered length adds, per member,
		 * len1 or what it reaches beyond the member before it (the first member counts in full) - a function of the member and its predecessor, so it
		 * sums in any order (modulo 2^29 like the reference's bit field).  Lane 0 alone walked the members with ~4 dependent reads each. */
		uint32_t j = 0;
		while(j < n_dst){
			const uint32_t gj = dv[j].gid;
			uint32_t i = j + 1;                                     /* end of the group: first member with another id */
			for(;;){
				const uint32_t y = i + lane;
				const unsigned long long m = __ballot(y >= n_dst || dv[y].gid != gj);
				if(m){ i += (uint32_t)__builtin_ctzll(m); break; }
				i += 64u;
			}
			int32_t b0 = 0x7FFFFFFF, b1 = 0x7FFFFFFF, e0 = 0, e1 = 0; uint32_t ov = 0;
			for(uint32_t k = j + lane; k < i; k += WTZ_NLANES){
				const wtz_zhit_t p = dv[k];
				const int32_t o1 = (int32_t)ZH_OFF1(p), l1 = (int32_t)ZH_LEN1(p), o2 = (int32_t)ZH_OFF2(p), l2 = (int32_t)ZH_LEN2(p);
				int32_t lst = 0;
				if(k > j){ const wtz_zhit_t pp = dv[k - 1]; lst = (int32_t)ZH_OFF1(pp) + (int32_t)ZH_LEN1(pp); }
				b0 = o1 < b0 ? o1 : b0; e0 = o1 + l1 > e0 ? o1 + l1 : e0;
				b1 = o2 < b1 ? o2 : b1; e1 = o2 + l2 > e1 ? o2 + l2 : e1;
				ov += (uint32_t)((o1 > lst) ? l1 : o1 + l1 - lst);
			}
			for(int d = 32; d > 0; d >>= 1){
				const int32_t xb0 = __shfl_xor(b0, d, 64), xb1 = __shfl_xor(b1, d, 64), xe0 = __shfl_xor(e0, d, 64), xe1 = __shfl_xor(e1, d, 64);
				ov += (uint32_t)__shfl_xor((int)ov, d, 64);
				b0 = xb0 < b0 ? xb0 : b0; b1 = xb1 < b1 ? xb1 : b1; e0 = xe0 > e0 ? xe0 : e0; e1 = xe1 > e1 ? xe1 : e1;
			}
			if(lane == 0){
				wtz_win_t seed;
				seed.pb2 = 0; seed.closed = 0; seed.dir = (uint8_t)dir; seed.pad = 0;
				seed.anchors[0] = j; seed.anchors[1] = i;
				seed.beg[0] = b0; seed.beg[1] = b1; seed.end[0] = e0; seed.end[1] = e1; seed.ovl = WTZ_OVL29(ov);
				if(!(seed.end[0] - seed.beg[0] < min_linear_len)) (void)S.regs[dir].push(seed);      /* a failed push sets the vector's `bad` flag (the caller reports the pool) */
			}
			j = i;
		}
	}
#else
	if(lane == 0){
		uint32_t j = 0;
		for(uint32_t i = 1; i <= n_dst; i++){
			if(i < n_dst && dv[i].gid == dv[j].gid) continue;
			wtz_win_t seed;
			seed.pb2 = 0; seed.closed = 0; seed.dir = (uint8_t)dir; seed.pad = 0;
			seed.anchors[0] = j; seed.anchors[1] = i;
			seed.beg[0] = seed.beg[1] = 0x7FFFFFFF; seed.end[0] = seed.end[1] = 0; seed.ovl = 0;
			int32_t lst = 0;
			for(uint32_t k = j; k < i; k++){
				const wtz_zhit_t p = dv[k];
				const int32_t o1 = (int32_t)ZH_OFF1(p), l1 = (int32_t)ZH_LEN1(p), o2 = (int32_t)ZH_OFF2(p), l2 = (int32_t)ZH_LEN2(p);
				if(o1 < seed.beg[0]) seed.beg[0] = o1;
				if(o1 + l1 > seed.end[0]) seed.end[0] = o1 + l1;
				if(o2 < seed.beg[1]) seed.beg[1] = o2;
				if(o2 + l2 > seed.end[1]) seed.end[1] = o2 + l2;
				seed.ovl = WTZ_OVL29(seed.ovl + (uint32_t)((o1 > lst) ? l1 : o1 + l1 - lst));
				lst = o1 + l1;
			}
			if(!(seed.end[0] - seed.beg[0] < min_linear_len)){ if(!S.regs[dir].push(seed)) break; }
			j = i;
		}
	}
#endif
	WTZ_WAVE_SYNC();
	WTZ_PROF_ADD(29, pdn);
	if(!IMG_LDS){ WTZ_PROF_ADD(10, pd0); WTZ_PROF_CNT(59, 1); }
	return 0;
#undef WTZ_GID
#undef WTZ_GID_SET
}

/* `big` (the last launch only): group ids of two bytes (up to WTZ_DM_GCAP_BIG linear groups instead of 255) and, when the per-match /
 * per-diagonal image still does not fit the slice next to the band work arrays, the image in the pool (the lane-0 loops then run
 * against L2 instead of LDS - several times slower than the LDS form, tens of times faster than the scalar body, which pays ~10
 * dependent HBM loads per match).  Returns 0 when done, else why not: 1 image too large, 2 band list overflow, 3 group table
 * overflow, 4 not applicable; nothing the caller sees has changed then. */
WTZ_HD int wtz_denoise_dir_coop(const wtz_zhit_t *rs, uint32_t n_rs, uint32_t dir, uint32_t nf, uint32_t nd, wtz_dmscratch_t &S, int32_t xvar, int32_t yvar, int32_t min_linear_len,
		uint8_t *lds, uint32_t lds_bytes, wtz_pool_t *pool, int32_t *bad, bool big = false){
	const uint32_t lane = WTZ_LANE;
	if(lds == NULL || n_rs > 65535u){ WTZ_PROF_CNT(60, 1000000); return 4; }
	const unsigned long long pd0 = WTZ_PROF_T(); (void)pd0;      /* 10 / 59: wave time / calls of the strands whose image lives in the pool */
	unsigned long long pdn = WTZ_PROF_T(); (void)pdn;      /* phase profiler: 24 image, 25 band list (lane 0), 26 productivity filter, 27 productive bands, 28 grouped order, 29 seeds (lane 0); 30 bands, 31 productive bands, 11 grouped matches */
	/* image: per match 4 B (off1<<10 | len1) + 1 B (group id; 2 B when big); per distinct diagonal 4 B (offset) + 2 B (first match) +
	 * 2 B (band members).  Work arrays: band keys / member lists and the group table.  The rs index of a match is only needed by the
	 * parallel passes: it lives in the pool. */
	/* members of one diagonal band the work arrays hold: the largest power of two (<= 2048) that fits beside this strand's image.  With a fixed 512 a
	 * quarter of the strands of configs[2] (1.8 M of 6.5 M calls: a true overlap puts hundreds of matches into one 64-diagonal band) overflowed it and were
	 * done again in the `big` form with their image in the pool. */
	uint32_t bcap = 2048u;
	if(!big) while(bcap > WTZ_DM_BCAP_MIN && wtz_dm_layout(nf, nd, 1u, 8u * bcap + 2u * WTZ_DM_GCAP + 40u).total > lds_bytes) bcap >>= 1;
	uint32_t gcap = WTZ_DM_GCAP;
	if(big){      /* the group table takes what the slice leaves beside the band arrays */
		if(lds_bytes < 8u * bcap + 40u + 2u * 512u) return 1;
		gcap = (lds_bytes - 8u * bcap - 40u) / 2u;
		if(gcap > WTZ_DM_GCAP_BIG) gcap = WTZ_DM_GCAP_BIG;
	}
	const uint32_t wrk_bytes = 8u * bcap + 2u * gcap + 40u;
	const wtz_dm_layout_t LY = wtz_dm_layout(nf, nd, big ? 2u : 1u, wrk_bytes);
	uint8_t *img = lds;
	if(!big){ if(LY.total > lds_bytes){ WTZ_PROF_CNT(61, 1000000); return 1; } }
	else if(wrk_bytes > lds_bytes) return 1;
	else if(LY.total > lds_bytes){
		const uint32_t dend = LY.off_g + 4u * (nd + 2u), img_bytes = LY.off_w > dend ? LY.off_w : dend;      /* in the pool the offsets overlay the group ids only */
		uint64_t ia = 0;
		if(lane == 0) ia = (uint64_t)(uintptr_t)wtz_pool_alloc(pool, (size_t)img_bytes + 16u);
		ia = wtz_coop_bcast64(ia);
		if(ia == 0){ *bad = 1; if(lane == 0) S.regs[dir].n = 0; return 0; }
		img = (uint8_t*)(uintptr_t)ia;
	}
	return (img == lds) ? wtz_denoise_dir_body<true>(rs, n_rs, dir, nf, nd, S, xvar, yvar, min_linear_len, lds, lds_bytes, pool, bad, big, bcap, gcap, LY, img, pd0, pdn)
	                    : wtz_denoise_dir_body<false>(rs, n_rs, dir, nf, nd, S, xvar, yvar, min_linear_len, lds, lds_bytes, pool, bad, big, bcap, gcap, LY, img, pd0, pdn);
}

/* ---------------------------------------------------------------------------------------------------------------------------------------------------
 * Block merging (fast_merge_wtseedv, hzm_aln.h:933-1054) as three steps over small arrays; executed by ONE lane (a strand has a handful of blocks - the
 * wave's share of this stage is the chain below).  What has to come out as the reference's does, and why it is written the way it is:
 *   - the blocks are first put in (diagonal, beg0) order and every block is a "diagonal" of its own (hzm_aln.h:942-955: `d` is reset per block), so a band
 *     is an index interval [lo, hi) of that order: hi stops at the first diagonal beyond D[lo] + span and never reaches the last block (957-962); a band
 *     whose closing diagonal equals that of the band emitted before is skipped whole (964-967); the next band starts at the first diagonal beyond
 *     D[lo] + span / 2 (1005-1008).  wtz_blockband walks that sequence over a plain array of diagonals.
 *   - inside a band the members are taken in beg0 order (the reference's own unstable sort: ties are observable) and cut into runs: a run is led by its
 *     first member and takes every following member that begins no later than the LEADER's end + xvar (982, 998: s0 only moves at a cut).
 *   - a run adopts the label of its first labelled member - through the forwarding table - and forwards the labels of its other labelled members to it
 *     (986-992); the table is then flattened by the reference's own three-level pass (1010-1019, not a full closure: kept as is), blocks are relabelled,
 *     put in (label, beg0) order, every label's blocks folded into the first one (bounding box, ovl summed in 29 bits) and the folded ones dropped.
 * --------------------------------------------------------------------------------------------------------------------------------------------------- */
struct wtz_blockband {
	const int32_t *D; uint32_t n; int32_t span;
	uint32_t at; int32_t closing; bool have_closing;
	WTZ_HDM void start(const int32_t *diag, uint32_t count, int32_t sp){ D = diag; n = count; span = sp; at = 0; closing = 0; have_closing = false; }
	/* the next band to process as [lo, hi); false when the sweep is over */
	WTZ_HDM bool next(uint32_t &lo, uint32_t &hi){
		while(at < n){
			const int32_t first = D[at];
			uint32_t len = 0;
			while(D[at + len] <= first + span && at + len + 1 < n) len++;
			if(len == 0) return false;
			const int32_t cl = D[at + len];
			if(have_closing && cl == closing){ at += len; continue; }
			closing = cl; have_closing = true;
			lo = at; hi = at + len;
			uint32_t nx = lo;
			while(nx < hi && D[nx] <= first + span / 2) nx++;
			at = nx;
			return true;
		}
		return false;
	}
};
/* label forwarding table of the merge: entry 0 = "no label" */
struct wtz_blocklabels {
	wtz_vec<uint32_t> *fw;
	WTZ_HDM bool reset(){ fw->n = 0; return fw->push(0u); }
	WTZ_HDM uint32_t fresh(){ const uint32_t g = fw->n; return fw->push(g) ? g : 0u; }
	/* one run of band members ord[a..b): all end up with the label the run adopts */
	WTZ_HDM bool adopt(wtz_win_t *blk, const uint32_t *ord, uint32_t a, uint32_t b){
		uint32_t lab = 0;
		for(uint32_t k = a; k < b; k++){
			const uint32_t old = blk[ord[k]].pb2;
			if(old == 0) continue;
			if(lab == 0) lab = fw->a[old]; else fw->a[old] = lab;
		}
		if(lab == 0){ lab = fresh(); if(lab == 0) return false; }
		for(uint32_t k = a; k < b; k++) blk[ord[k]].pb2 = lab;
		return true;
	}
};

WTZ_HD void wtz_merge_blocks(wtz_vec<wtz_win_t> &rv, wtz_dmscratch_t &S, int32_t xvar, int32_t yvar){
	wtz_win_t *blk = rv.a; const uint32_t n = rv.n;
	wtz_sort_exact(blk, (size_t)n, wtz_gt_wdiag());
	/* the diagonals as a plain array (in the `offset` fields of the diagonal scratch: its other fields are not needed - every block is its own diagonal) */
	S.diags.n = 0;
	if(!S.diags.reserve(n + 2)) return;
	int32_t *D = (int32_t*)S.diags.a;
	for(uint32_t i = 0; i < n; i++) D[i] = blk[i].beg[0] - blk[i].beg[1];
	wtz_blocklabels labels; labels.fw = &S.grps;
	if(!labels.reset()) return;
	wtz_blockband sweep; sweep.start(D, n, yvar);
	uint32_t lo, hi;
	while(sweep.next(lo, hi)){
		S.block.n = 0;
		if(!S.block.reserve(hi - lo)) return;
		uint32_t *ord = S.block.a; const uint32_t m = hi - lo;
		for(uint32_t k = 0; k < m; k++) ord[k] = lo + k;
		S.block.n = m;
		wtz_gt_widx_beg0 by_beg; by_beg.rs = blk;
		wtz_sort_exact(ord, (size_t)m, by_beg);
		uint32_t lead = 0;
		for(uint32_t k = 1; k <= m; k++){
			if(k < m && blk[ord[k]].beg[0] <= blk[ord[lead]].end[0] + xvar) continue;      /* still inside the leader's reach (k == m: the sentinel of hzm_aln.h:940 starts beyond everything) */
			if(!labels.adopt(blk, ord, lead, k)) return;
			lead = k;
		}
	}
	wtz_tidy_groups(S.grps.a, S.grps.n);
	for(uint32_t i = 0; i < n; i++) if(blk[i].pb2) blk[i].pb2 = S.grps.a[blk[i].pb2];
	wtz_sort_exact(blk, (size_t)n, wtz_gt_wgrp());
	uint32_t head = 0;
	while(head < n && blk[head].pb2 == 0) head++;                   /* unlabelled blocks stay as they are (they sort first) */
	while(head < n){
		uint32_t tail = head + 1;
		wtz_win_t acc = blk[head];
		for(; tail < n && blk[tail].pb2 == acc.pb2; tail++){
			const wtz_win_t &o = blk[tail];
			acc.beg[0] = o.beg[0] < acc.beg[0] ? o.beg[0] : acc.beg[0]; acc.end[0] = o.end[0] > acc.end[0] ? o.end[0] : acc.end[0];
			acc.beg[1] = o.beg[1] < acc.beg[1] ? o.beg[1] : acc.beg[1]; acc.end[1] = o.end[1] > acc.end[1] ? o.end[1] : acc.end[1];
			acc.ovl = WTZ_OVL29(acc.ovl + o.ovl);
			blk[tail].closed = 1;
		}
		blk[head] = acc;
		head = tail;
	}
	wtz_sort_exact(blk, (size_t)n, wtz_gt_wclosed());
	uint32_t kept = 0;
	while(kept < n && !blk[kept].closed) kept++;
	rv.n = kept;
}

WTZ_HD int32_t wtz_w30(int32_t v){ return (int32_t)((uint32_t)v << 2) >> 2; }     /* node_t.weight:30, hzm_aln.h:1057 */

/* ---------------------------------------------------------------------------------------------------------------------------------------------------
 * Chain of blocks (chaining_overhang_wtseedv, hzm_aln.h:1056-1132) on the whole wavefront.  The reference runs  for i: for j > i  with a `continue` for
 * blocks that start too far before i's end and a `break` at the first block (not skipped) that starts more than W = weight(i) / gap_penalty beyond it.
 * The blocks are in beg0 order, so "beyond W" is monotone in j: once a block is, every later one is too - and a later block that the `continue`s would have
 * skipped changes nothing either way.  The update of j from i is therefore a LOCAL decision (not skipped, within W), independent of the other j: the inner
 * loop is one step of 64 lanes (more blocks: further steps, ended by the same monotone test), and only the outer loop is sequential - n steps instead of
 * n^2 / 2.  State per block in `st` (3 n words: weight, predecessor, head flag).  Entered by every lane with uniform arguments; blk / st may be LDS or pool.
 * Host emulation: one lane.
 * --------------------------------------------------------------------------------------------------------------------------------------------------- */
WTZ_HD int32_t wtz_chain_blocks_coop(int32_t len0, int32_t len1, wtz_win_t *blk, uint32_t n, int32_t *st, int32_t margin, int32_t overhang, float dev_penalty, float gap_penalty){
	const uint32_t lane = WTZ_LANE;
	int32_t *wt = st, *from = st + n, *hd = st + 2 * n;
	for(uint32_t j = lane; j < n; j += WTZ_NLANES){
		wt[j] = 0; from[j] = -1;
		hd[j] = (blk[j].beg[0] <= margin || blk[j].beg[1] <= margin) ? 1 : 0;
		blk[j].closed = 1;
	}
	WTZ_WAVE_SYNC();
	int32_t best = -1000000, best_at = -1;
	for(uint32_t i = 0; i < n; i++){
		const int32_t e0 = blk[i].end[0], e1 = blk[i].end[1];
		const int32_t wi = wtz_w30(wt[i] + (int32_t)blk[i].ovl), hi = hd[i];
		const int32_t ti = (e0 + margin > len0 || e1 + margin > len1) ? 1 : 0;
		const int32_t rank = wi * ((hi + 3) * (ti + 3)) / 16;
		if(rank > best){ best = rank; best_at = (int32_t)i; }
		const int32_t reach = (int32_t)((float)wi / gap_penalty);
		WTZ_WAVE_SYNC();                                              /* every lane has read wt[i] before lane 0 rewrites it */
		if(lane == 0) wt[i] = wi;
		for(uint32_t j0 = i + 1; j0 < n; j0 += WTZ_NLANES){
			const uint32_t j = j0 + lane;
			bool past = false;
			if(j < n){
				const int32_t b0 = blk[j].beg[0], b1 = blk[j].beg[1];
				const bool skipped = (b0 + overhang < e0) || (b1 + overhang < e1);
				const int32_t d0 = b0 - e0, d1 = b1 - e1;
				if(!skipped){
					if(d0 > reach) past = true;
					else {
						const int32_t band = WTZ_ABSDIFF(d0, d1);
						int32_t gap = WTZ_MAX(d0, d1); if(gap < 0) gap = -gap;
						const float fa = (float)band * dev_penalty, fb = (float)gap * gap_penalty;
						const int32_t sc = wi - (int32_t)(fa + fb);
						if(wt[j] <= sc){ wt[j] = wtz_w30(sc); from[j] = (int32_t)i; hd[j] = hi; }
					}
				}
			}
			if(wtz_coop_ballot(past) != 0ull) break;
		}
		WTZ_WAVE_SYNC();
	}
	int32_t total = 0;
	for(int32_t k = best_at; k >= 0; k = from[k]){ if(lane == 0) blk[k].closed = 0; total += (int32_t)blk[k].ovl; }
	WTZ_WAVE_SYNC();
	return total;
}

/* Entered by every lane of the wavefront (lane 0 alone in the host emulation); the result is valid on lane 0.
 * `lds`: the wave's LDS slice - the strand images of the denoise pass first, then the handful of blocks and the scratch
 * vectors of block merging / chaining (they are chains of tiny order-sensitive sorts: latency, not bandwidth). */
WTZ_HD wtz_dm_result_t wtz_dot_matrix_align(wtz_vec<wtz_zhit_t> &cache, wtz_pool_t *pool, int32_t pblen1, int32_t pblen2, const wtz_params_t *P, int32_t *bad, bool presorted, uint64_t *tick_denoise,
		uint8_t *lds, uint32_t lds_bytes, bool defer_if_large, bool allow_big){
	wtz_dm_result_t ret; int32_t weight[2]; uint32_t d;
	memset(&ret, 0, sizeof ret);
	const uint32_t lane = WTZ_LANE;
	if(!presorted){      /* hzm_aln.h:728 on lane 0 (the wave-wide order had an observable tie and no LDS room to replay it) */
		if(lane == 0) wtz_sort_exact(cache.a, (size_t)cache.n, wtz_gt_zdiag());
		WTZ_WAVE_SYNC();
		presorted = true;
	}
	wtz_dm_counts_t cnts; cnts.nf[0] = cnts.nf[1] = cnts.nd[0] = cnts.nd[1] = 0;
	if(cache.n <= 65535u && lds){
		cnts = wtz_dm_counts(cache.a, cache.n);
		const uint32_t need0 = wtz_denoise_lds_need(cnts.nf[0], cnts.nd[0], lds_bytes), need1 = wtz_denoise_lds_need(cnts.nf[1], cnts.nd[1], lds_bytes);
		if(defer_if_large && !allow_big && (need0 > need1 ? need0 : need1) > lds_bytes){ ret.dir = -2; ret.score = (int32_t)(need0 > need1 ? need0 : need1); return ret; }      /* a later launch with a larger LDS slice takes the pair */
	}
	wtz_dmscratch_t S;
	S.dst.a = NULL; S.dst.n = S.dst.cap = 0; S.dst.pool = pool; S.dst.bad = 0;
	S.regs[0] = S.regs[1] = wtz_vec<wtz_win_t>(); S.regs[0].a = S.regs[1].a = NULL; S.regs[0].n = S.regs[0].cap = S.regs[1].n = S.regs[1].cap = 0; S.regs[0].pool = S.regs[1].pool = pool; S.regs[0].bad = S.regs[1].bad = 0;
	S.diags.a = NULL; S.diags.n = S.diags.cap = 0; S.diags.pool = pool; S.diags.bad = 0;
	S.block.a = NULL; S.block.n = S.block.cap = 0; S.block.pool = pool; S.block.bad = 0;
	S.grps.a = NULL; S.grps.n = S.grps.cap = 0; S.grps.pool = pool; S.grps.bad = 0;
	if(lane == 0){
		S.dst.init(pool, cache.n / 2 + 16); S.regs[0].init(pool, 16); S.regs[1].init(pool, 16);
		S.diags.init(pool, 64); S.block.init(pool, 64); S.grps.init(pool, 16);
	}
	WTZ_WAVE_SYNC();
	/* dst of strand 0 is consumed before strand 1 reuses it: regs only keep bounds */
	for(uint32_t dir = 0; dir < 2; dir++){
		const unsigned long long ptd = WTZ_PROF_T();
		int why = 4;
		if(cache.n <= 65535u && lds){
			/* the strand image in LDS when it fits the slice, else (allow_big) in the pool with wide group ids: the pair stays in this launch */
			const bool fits = wtz_denoise_lds_need(cnts.nf[dir], cnts.nd[dir], lds_bytes) <= lds_bytes;
			why = wtz_denoise_dir_coop(cache.a, cache.n, dir, cnts.nf[dir], cnts.nd[dir], S, P->xvar, P->yvar, P->min_block_len, lds, lds_bytes, pool, bad, allow_big && !fits);
			if(why && allow_big && fits) why = wtz_denoise_dir_coop(cache.a, cache.n, dir, cnts.nf[dir], cnts.nd[dir], S, P->xvar, P->yvar, P->min_block_len, lds, lds_bytes, pool, bad, true);      /* more than 255 groups, or a band beyond the small slice's member list */
			if(why && defer_if_large){ ret.dir = -2; ret.score = 0; return ret; }      /* band / group table overflow: the next launch takes the pair */
		}
		if(why){
			const unsigned long long ptf = WTZ_PROF_T();
			if(lane == 0) wtz_denoise_dir(cache.a, cache.n, dir, S, P->xvar, P->yvar, P->min_block_len);
			WTZ_WAVE_SYNC();
			WTZ_PROF_ADD(3, ptf); WTZ_PROF_CNT(1, 1); WTZ_PROF_CNT(5, cache.n);
		} else { WTZ_PROF_ADD(2, ptd); WTZ_PROF_CNT(0, 1); WTZ_PROF_CNT(4, cache.n); }
	}
#if defined(__HIP_DEVICE_COMPILE__)
	*tick_denoise = (uint64_t)clock64();
#else
	*tick_denoise = 0;
#endif
	/* blocks + scratch of the merge pass into LDS when they are few (the usual case); lane 0 merges, the wave chains */
	const unsigned long long ptm = WTZ_PROF_T();
	int32_t *chain_mem = NULL;
	if(lane == 0){
		if(lds && lds_bytes >= 12288u && S.regs[0].n <= 64u && S.regs[1].n <= 64u){
			wtz_win_t *r0 = (wtz_win_t*)lds, *r1 = r0 + 64;
			for(uint32_t i = 0; i < S.regs[0].n; i++) r0[i] = S.regs[0].a[i];
			for(uint32_t i = 0; i < S.regs[1].n; i++) r1[i] = S.regs[1].a[i];
			S.regs[0].a = r0; S.regs[0].cap = 64; S.regs[1].a = r1; S.regs[1].cap = 64;
			uint8_t *q = (uint8_t*)(r1 + 64);
			S.diags.a = (wtz_diag_t*)q; S.diags.n = 0; S.diags.cap = 72; q += 72 * sizeof(wtz_diag_t);
			S.block.a = (uint32_t*)q; S.block.n = 0; S.block.cap = 128; q += 128 * 4;
			S.grps.a = (uint32_t*)q; S.grps.n = 0; S.grps.cap = 128; q += 128 * 4;
			chain_mem = (int32_t*)q;                                   /* 3 x 64 words */
		}
		wtz_merge_blocks(S.regs[0], S, P->xvar, 2 * P->yvar);
		wtz_merge_blocks(S.regs[1], S, P->xvar, 2 * P->yvar);
		for(int k = 0; k < 2; k++) wtz_sort_exact(S.regs[k].a, (size_t)S.regs[k].n, wtz_gt_wbeg0());      /* hzm_aln.h:1065 */
		if(chain_mem == NULL || S.regs[0].n > 64u || S.regs[1].n > 64u){
			const uint32_t m = S.regs[0].n > S.regs[1].n ? S.regs[0].n : S.regs[1].n;
			chain_mem = (int32_t*)wtz_pool_alloc(pool, (size_t)(3u * m + 4u) * 4u);
			if(chain_mem == NULL) *bad = 1;
		}
	}
	WTZ_WAVE_SYNC();
	wtz_win_t *cb[2]; uint32_t cn[2];
	for(int k = 0; k < 2; k++){ cb[k] = (wtz_win_t*)(uintptr_t)wtz_coop_bcast64((uint64_t)(uintptr_t)S.regs[k].a); cn[k] = wtz_coop_bcast32(S.regs[k].n); }
	chain_mem = (int32_t*)(uintptr_t)wtz_coop_bcast64((uint64_t)(uintptr_t)chain_mem);
	if(chain_mem == NULL){ weight[0] = weight[1] = 0; cn[0] = cn[1] = 0; }
	else for(int k = 0; k < 2; k++) weight[k] = wtz_chain_blocks_coop(pblen1, pblen2, cb[k], cn[k], chain_mem, P->xvar, P->max_overhang, P->deviation_penalty, P->gap_penalty);
	WTZ_PROF_ADD(6, ptm);
	if(lane != 0) return ret;
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
