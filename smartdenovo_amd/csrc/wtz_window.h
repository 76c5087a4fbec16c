/*
 * wtz_window.h — per-pair task bodies: z-mer matching (A6), the (off1,off2) ordering, window
 * detection and window chaining of the zmo engine (A7z).
 *
 *   wtz_zmatch            hzm_aln.h:173-224   query_single_read_seeds against the prebuilt tables
 *   wtz_scan_windows_coop hzm_aln.h:410-578   potential_paired_kmers_windows (fast chaining branch; its median, hzm_aln.h:316-343, is a rank selection on the sorting network)
 *   wtz_merge_windows_wave hzm_aln.h:580-656  merge_paired_kmers_window
 *   wtz_chain_windows_wave hzm_aln.h:658-713  chaining_wtseedv
 * Every body is wave-cooperative (one lane in the host emulation); round 6 removed the scalar restatements that used to sit behind "does not fit" branches
 * (a range that fits neither the LDS slice nor a pool workspace means the pool is exhausted: the stage ends in WTZ_E_POOL and the host redoes the range).
 *
 * All ordering-sensitive steps use wtz_sort_exact (the reference's unstable sort) because
 * tie order leaks into the output (SURVEY §8a trap 1).  Integer promotions follow the
 * reference's bit-fields: offsets/lengths are non-negative and mix with uint32 as unsigned.
 */
#ifndef WTZ_WINDOW_H
#define WTZ_WINDOW_H

#include "wtz_seed.h"

#define WTZ_KWIN_MAX_OFFSET_DEV 50

/* ---- A6: walk the candidate's position-ordered z-mers, look each up in the query's table ---- */
WTZ_HD bool wtz_zmatch(const wtz_zindex_t &ZQ, const wtz_zindex_t &ZC, uint32_t q, uint32_t c, uint32_t clen, uint32_t max_var, wtz_vec<wtz_zhit_t> &out, bool same_strand = false){
	const uint64_t qo = ZQ.zoff[q], co = ZC.zoff[c];      /* ZQ: the index holding the query's table (A5), ZC: the one holding the candidate's position-ordered z-mers (the same index unless the caller split them) */
	const uint32_t cn = (uint32_t)(ZC.zoff[c + 1] - co), qd = ZQ.dn[q];
	const uint32_t *dmer = ZQ.dmer + qo;
	for(uint32_t k = 0; k < cn; k++){
		if(!ZC.ok[co + k]) continue;                       /* per-table-entry hit cap (hzm_aln.h:208-211) */
		uint32_t m = ZC.mer[co + k];
		uint32_t lo = 0, hi = qd;
		while(lo < hi){ uint32_t mid = (lo + hi) >> 1; if(dmer[mid] < m) lo = mid + 1; else hi = mid; }
		if(lo >= qd || dmer[lo] != m) continue;
		uint32_t cpos = ZC.pos[co + k], clen2 = ZC.len[co + k];
		uint32_t first = ZQ.dfirst[qo + lo], cnt = ZQ.dcnt[qo + lo];
		for(uint32_t e = 0; e < cnt; e++){
			uint32_t qi = ZQ.sidx[qo + first + e];
			uint32_t qpos = ZQ.pos[qo + qi], qlen = ZQ.len[qo + qi];
			uint32_t dv = qlen > clen2 ? qlen - clen2 : clen2 - qlen;
			if(dv > max_var) continue;
			uint32_t d1 = qpos & 1u, d2 = cpos & 1u;
			if(same_strand && (d1 ^ d2)) continue;           /* align_hzmaux: filter_by_region_hzmps(dir 0), hzm_aln.h:1193 */
			uint32_t off2 = (d1 ^ d2) ? clen - ((cpos >> 1) + clen2) : (cpos >> 1);
			wtz_zhit_t h; h.o1 = (d1 << 31) | (qpos >> 1); h.o2 = (d2 << 31) | off2; h.ll = (clen2 << 16) | qlen; h.gid = 0;
			if(!out.push(h)) return false;
		}
	}
	return true;
}

/* A6, wave-cooperative: lanes take consecutive candidate z-mers, the matches of a 64-z-mer chunk are written in
 * candidate-position order through an exclusive scan of the per-lane match counts (emission order is part of the
 * contract: hzm_aln.h:212-221).  Passes: (round 4) prefilter + ordered compaction, count (the dense index of each surviving z-mer is
 * cached), allocate exactly, fill.
 * The prefilter: of a candidate's ~7 500 homopolymer-compressed 10-mers only ~10 % occur in the query at all (4 * 3^9 = 78 732 such z-mers,
 * ~7 500 distinct in a 10 kb read), yet every one of them paid a 13-step binary search of dependent L2 loads in the query's table - with
 * 64 lanes in lockstep a chunk costs the full chain whenever ANY lane has work, so skipping per lane buys nothing.  Instead the wave first
 * sets one bit per distinct query z-mer in a bitmap in its LDS slice (multiplicative hash), streams the candidate's z-mers past it and
 * COMPACTS the ones whose bit is set (ballot ranks: order kept); only those - a fifth - are searched.  A false positive is searched and
 * found absent like before; the emitted matches and their order are unchanged.  `lds` == NULL (host emulation) or a query too long for
 * the bitmap: every candidate z-mer is a survivor.
 * Returns the match list through *out / *n_out on every lane; false when the pool is exhausted. */
WTZ_HD bool wtz_zmatch_coop(const wtz_zindex_t &ZQ, const wtz_zindex_t &ZC, uint32_t q, uint32_t c, uint32_t clen, uint32_t max_var, wtz_pool_t *pool, wtz_zhit_t **out, uint32_t *n_out, bool same_strand = false,
		uint32_t *lds = NULL, uint32_t lds_bytes = 0){
	const uint64_t qo = ZQ.zoff[q], co = ZC.zoff[c];      /* ZQ: the index holding the query's table (A5), ZC: the one holding the candidate's position-ordered z-mers (the same index unless the caller split them) */
	const uint32_t cn = (uint32_t)(ZC.zoff[c + 1] - co), qd = ZQ.dn[q];
	const uint32_t *dmer = ZQ.dmer + qo;
	const uint32_t lane = WTZ_LANE;
	uint64_t pa = 0;
	if(lane == 0) pa = (uint64_t)(uintptr_t)wtz_pool_alloc(pool, (size_t)(cn + 1) * 12);
	pa = wtz_coop_bcast64(pa);
	uint32_t *found = (uint32_t*)(uintptr_t)pa;
	if(found == NULL) return false;
	uint32_t *kcnt = found + (cn + 1), *surv = kcnt + (cn + 1);
	/* ---- prefilter ---- */
	uint32_t lb = 0;                                          /* log2 of the bitmap's bits; 0 = no filter */
	if(lds != NULL && lds_bytes >= 2048u && cn >= 256u){
		lb = 14; while(lb < 20u && ((2u << lb) >> 3) <= lds_bytes) lb++;
		if((uint64_t)qd * 3u > (1ull << lb)) lb = 0;            /* more than a third of the bits would be set: not worth a pass */
	}
	if(lb){
		const uint32_t nw = (1u << lb) >> 5;
		for(uint32_t w = lane; w < nw; w += WTZ_NLANES) lds[w] = 0u;
		WTZ_WAVE_SYNC();
		for(uint32_t i = lane; i < qd; i += WTZ_NLANES){
			const uint32_t h = (dmer[i] * 0x9E3779B1u) >> (32u - lb);
#if defined(__HIP_DEVICE_COMPILE__)
			atomicOr(&lds[h >> 5], 1u << (h & 31u));
#else
			lds[h >> 5] |= 1u << (h & 31u);
#endif
		}
		WTZ_WAVE_SYNC();
	}
	uint32_t ns = 0;
	for(uint32_t k0 = 0; k0 < cn; k0 += 4 * WTZ_NLANES){
		uint32_t mm[4]; bool act[4];
		#pragma unroll
		for(int u = 0; u < 4; u++){
			const uint32_t k = k0 + u * WTZ_NLANES + lane;
			act[u] = k < cn && ZC.ok[co + k];              /* per-table-entry hit cap (hzm_aln.h:208-211) */
			mm[u] = act[u] ? ZC.mer[co + k] : 0u;
		}
		#pragma unroll
		for(int u = 0; u < 4; u++){
			const uint32_t k = k0 + u * WTZ_NLANES + lane;
			bool keep = act[u];
			if(lb && keep){ const uint32_t h = (mm[u] * 0x9E3779B1u) >> (32u - lb); keep = ((lds[h >> 5] >> (h & 31u)) & 1u) != 0u; }
			uint32_t tot; const uint32_t pos = wtz_coop_rank(keep, &tot);
			if(keep) surv[ns + pos] = k;
			ns += tot;
		}
	}
	WTZ_WAVE_SYNC();
	/* ---- the surviving z-mers in the query's table: dense index + number of query occurrences.  Four survivors per lane and iteration: the binary
	 * searches are chains of dependent loads, so four independent chains in flight per lane (stepped in lockstep) hide most of their latency ---- */
	uint32_t nflat = 0;                                       /* (candidate z-mer, query occurrence) combinations, before the length / strand tests */
	for(uint32_t s0 = 0; s0 < ns; s0 += 4 * WTZ_NLANES){
		uint32_t m[4], lo[4], hi[4]; bool act[4];
		#pragma unroll
		for(int u = 0; u < 4; u++){
			const uint32_t si = s0 + u * WTZ_NLANES + lane;
			act[u] = si < ns;
			m[u] = act[u] ? ZC.mer[co + surv[si]] : 0u;
			lo[u] = 0; hi[u] = act[u] ? qd : 0u;
		}
		for(;;){
			bool any = false;
			uint32_t mid[4], dv[4];
			#pragma unroll
			for(int u = 0; u < 4; u++){ mid[u] = (lo[u] + hi[u]) >> 1; dv[u] = (lo[u] < hi[u]) ? dmer[mid[u]] : 0u; }
			#pragma unroll
			for(int u = 0; u < 4; u++){ if(lo[u] < hi[u]){ if(dv[u] < m[u]) lo[u] = mid[u] + 1; else hi[u] = mid[u]; any = any || (lo[u] < hi[u]); } }
			if(!any) break;
		}
		#pragma unroll
		for(int u = 0; u < 4; u++){
			const uint32_t si = s0 + u * WTZ_NLANES + lane;
			uint32_t cnt = 0, first = 0;
			if(act[u] && lo[u] < qd && dmer[lo[u]] == m[u]){ first = ZQ.dfirst[qo + lo[u]]; cnt = ZQ.dcnt[qo + lo[u]]; }
			uint32_t chunk; const uint32_t o = nflat + wtz_coop_excl_scan(cnt, &chunk);
			if(si < ns){ found[si] = first; kcnt[si] = o; }          /* found: first entry of the z-mer's occurrence list; kcnt: where its combinations start in the flat order */
			nflat += chunk;
		}
	}
	/* ---- Round 4: the combinations FLAT, one per lane.  A lane used to walk its z-mer's occurrence list by itself (up to -Z of them, three dependent loads
	 * each, twice: count, then fill) while the other lanes of the chunk waited for the longest list; the flat order (candidate position, then occurrence:
	 * hzm_aln.h:212-221) is the emission order, so one pass with an ordered compaction of the combinations that pass the length / strand tests writes the
	 * matches directly.  The list of (survivor, occurrence) is materialised first: a store per combination, no loads. ---- */
	pa = 0;
	if(lane == 0) pa = (uint64_t)(uintptr_t)wtz_pool_alloc(pool, (size_t)(nflat + 1) * 8 + (size_t)(nflat + 2) * sizeof(wtz_zhit_t));
	pa = wtz_coop_bcast64(pa);
	if(pa == 0) return false;
	wtz_zhit_t *hits = (wtz_zhit_t*)(uintptr_t)pa;                     /* room for every combination; `total` of them are used */
	uint32_t *fs = (uint32_t*)(hits + (nflat + 2)), *fq = fs + (nflat + 1);
	WTZ_WAVE_SYNC();                                                  /* kcnt[] / found[] of the neighbouring lanes */
	for(uint32_t s0 = 0; s0 < ns; s0 += WTZ_NLANES){
		const uint32_t si = s0 + lane;
		if(si < ns){
			const uint32_t o = kcnt[si], e1 = (si + 1 < ns) ? kcnt[si + 1] : nflat, first = found[si];
			for(uint32_t e = o; e < e1; e++){ fs[e] = si; fq[e] = first + (e - o); }
		}
	}
	WTZ_WAVE_SYNC();
	uint32_t total = 0;
	for(uint32_t t0 = 0; t0 < nflat; t0 += 4 * WTZ_NLANES){
		uint32_t kq[4], qi[4], qpos[4], qlen[4], cpos[4], clen2[4]; bool in[4];
		#pragma unroll
		for(int u = 0; u < 4; u++){ const uint32_t t = t0 + u * WTZ_NLANES + lane; in[u] = t < nflat; kq[u] = in[u] ? surv[fs[t]] : 0u; qi[u] = in[u] ? ZQ.sidx[qo + fq[t]] : 0u; }
		#pragma unroll
		for(int u = 0; u < 4; u++){
			qpos[u] = in[u] ? ZQ.pos[qo + qi[u]] : 0u; qlen[u] = in[u] ? ZQ.len[qo + qi[u]] : 0u;
			cpos[u] = in[u] ? ZC.pos[co + kq[u]] : 0u; clen2[u] = in[u] ? ZC.len[co + kq[u]] : 0u;
		}
		#pragma unroll
		for(int u = 0; u < 4; u++){
			const uint32_t dv = qlen[u] > clen2[u] ? qlen[u] - clen2[u] : clen2[u] - qlen[u];
			const uint32_t d1 = qpos[u] & 1u, d2 = cpos[u] & 1u;
			const bool keep = in[u] && dv <= max_var && !(same_strand && (d1 ^ d2));      /* |len1 - len2| <= -l (hzm_aln.h:214); align_hzmaux keeps strand 0 only (hzm_aln.h:1193) */
			uint32_t tot; const uint32_t pos = wtz_coop_rank(keep, &tot);
			if(keep){
				const uint32_t off2 = (d1 ^ d2) ? clen - ((cpos[u] >> 1) + clen2[u]) : (cpos[u] >> 1);
				wtz_zhit_t h; h.o1 = (d1 << 31) | (qpos[u] >> 1); h.o2 = (d2 << 31) | off2; h.ll = (clen2[u] << 16) | qlen[u]; h.gid = 0;
				hits[total + pos] = h;
			}
			total += tot;
		}
	}
	if(lane == 0){ wtz_zhit_t z0; z0.o1 = z0.o2 = z0.ll = z0.gid = 0; hits[total] = z0; hits[total + 1] = z0; }   /* the element the reference reads past the end */
	*out = hits; *n_out = total;
	return true;
}

struct wtz_gt_off12 { WTZ_HDM bool operator()(const wtz_zhit_t &a, const wtz_zhit_t &b) const {
	uint64_t ka = ((uint64_t)ZH_OFF1(a) << 32) | ZH_OFF2(a), kb = ((uint64_t)ZH_OFF1(b) << 32) | ZH_OFF2(b); return ka > kb; } };
struct wtz_gt_off1 { WTZ_HDM bool operator()(const wtz_zhit_t &a, const wtz_zhit_t &b) const { return ZH_OFF1(a) > ZH_OFF1(b); } };
struct wtz_gt_idx_off2 { const wtz_zhit_t *rs; WTZ_HDM bool operator()(uint32_t a, uint32_t b) const { return ZH_OFF2(rs[a]) > ZH_OFF2(rs[b]); } };

#if defined(__HIP_DEVICE_COMPILE__)
/*
 * process_hzmps (hzm_aln.h:1184-1186) on the whole wavefront.  The reference orders the matches by (off1,off2) with its
 * UNSTABLE sort; only exact key ties (cross-strand coincidences) make the result depend on the swap sequence.  So: bitonic
 * sort of packed words off1:24 | off2:24 | index:16 by all 64 lanes (LDS when it fits, else the pool), then a tie scan -
 * no ties means the ascending order is unique and equals the reference's; any tie makes the caller run the swap-exact
 * sequential sort instead.  Returns the sorted copy (with the two zeroed sentinels) or NULL (tie / too large / pool).
 */
template<int SH> struct wtz_gt_keybits { WTZ_HDM bool operator()(uint64_t a, uint64_t b) const { return (a >> SH) > (b >> SH); } };
/* the part of wtz_sort_hits_wave behind the choice of where the key words live, compiled for either answer: with `w = fits ? lds : pool` in one body every
 * access to the keys was a FLAT instruction (see wtz_denoise_dir_body) */
template<int BYDIAG, bool W_LDS>
WTZ_D wtz_zhit_t *wtz_sort_hits_body(const wtz_zhit_t *hits, uint32_t n, wtz_pool_t *pool, uint64_t *lds, uint32_t lds_u64, uint64_t *w_pool, uint32_t np, int *pool_bad){
	const uint32_t lane = WTZ_LANE;
	const int SH = BYDIAG ? 15 : 16;             /* index bits below the key */
	uint64_t *w = W_LDS ? lds : w_pool;
	for(uint32_t i = lane; i < np; i += 64){
		uint64_t v = ~0ull;
		if(i < n){
			if(BYDIAG) v = ((uint64_t)((int64_t)ZH_OFF1(hits[i]) - (int64_t)ZH_OFF2(hits[i]) + (1 << 24)) << 39) | ((uint64_t)ZH_OFF1(hits[i]) << 15) | i;   /* denoising_hzmps order, hzm_aln.h:728 */
			else       v = ((uint64_t)ZH_OFF1(hits[i]) << 40) | ((uint64_t)ZH_OFF2(hits[i]) << 16) | i;                                                              /* process_hzmps order, hzm_aln.h:1185 */
		}
		w[i] = v;
	}
	__threadfence_block();
	if(W_LDS) wtz_coop_sort_u64(w, np);
	else {       /* key words in HBM: sort through the LDS window (largest power of two that fits) */
		uint32_t ln = 128; while(lds && ln * 2 <= lds_u64) ln <<= 1;
		wtz_coop_sort_u64_windowed(w, np, lds, lds ? ln : 0);
	}
	/* equal keys: the swap sequence of the reference decides - except, for the (off1,off2) order of the zmo engine, when the
	 * tied matches also agree in len1.  Those are cross-strand twins of ONE query z-mer; merge_paired_kmers_window reads
	 * other-strand matches only through (off1, len1) (hzm_aln.h:615-650) and filters by strand everywhere else, so their
	 * relative order cannot be observed (runs of three or more equal keys still take the exact path) */
	uint32_t tie = 0;
	for(uint32_t i = lane; i + 1 < n; i += 64) if((w[i] >> SH) == (w[i + 1] >> SH)){
		if(BYDIAG) tie = 1;
		else { const wtz_zhit_t &a = hits[(uint32_t)(w[i] & 0xFFFFu)], &b = hits[(uint32_t)(w[i + 1] & 0xFFFFu)]; if(ZH_LEN1(a) != ZH_LEN1(b) || ZH_STRAND(a) == ZH_STRAND(b) || (i + 2 < n && (w[i] >> SH) == (w[i + 2] >> SH))) tie = 1; }
	}
	uint32_t any; (void)wtz_coop_excl_scan(tie, &any);
	if(any){
		/* an observable tie: the reference's swap sequence decides.  When the key words sit in LDS, lane 0 replays it there
		 * on (key | index) words from the ORIGINAL order - the comparator looks at the key bits only, so the swaps are those of
		 * sorting the matches themselves - instead of chasing 16-byte records through HBM */
		if(!W_LDS) return NULL;
		for(uint32_t i = lane; i < n; i += 64){
			uint64_t v;
			if(BYDIAG) v = ((uint64_t)((int64_t)ZH_OFF1(hits[i]) - (int64_t)ZH_OFF2(hits[i]) + (1 << 24)) << 39) | ((uint64_t)ZH_OFF1(hits[i]) << 15) | i;
			else       v = ((uint64_t)ZH_OFF1(hits[i]) << 40) | ((uint64_t)ZH_OFF2(hits[i]) << 16) | i;
			w[i] = v;
		}
		__threadfence_block();
		if(lane == 0){ if(BYDIAG) wtz_sort_exact(w, (size_t)n, wtz_gt_keybits<15>()); else wtz_sort_exact(w, (size_t)n, wtz_gt_keybits<16>()); }
		__threadfence_block();
	}
	uint64_t oa = 0;
	if(lane == 0) oa = (uint64_t)(uintptr_t)wtz_pool_alloc(pool, (size_t)(n + 2) * sizeof(wtz_zhit_t));
	oa = wtz_coop_bcast64(oa);
	wtz_zhit_t *out = (wtz_zhit_t*)(uintptr_t)oa;
	if(out == NULL){ *pool_bad = 1; return NULL; }
	for(uint32_t i = lane; i < n; i += 64) out[i] = hits[(uint32_t)(w[i] & ((1u << SH) - 1u))];
	if(lane == 0){ wtz_zhit_t z0; z0.o1 = z0.o2 = z0.ll = z0.gid = 0; out[n] = z0; out[n + 1] = z0; }
	__threadfence_block();
	return out;
}
template<int BYDIAG>
WTZ_D wtz_zhit_t *wtz_sort_hits_wave(const wtz_zhit_t *hits, uint32_t n, wtz_pool_t *pool, uint64_t *lds, uint32_t lds_u64, int *pool_bad){
	const uint32_t lane = WTZ_LANE;
	*pool_bad = 0;
	if(n < 2 || n > (BYDIAG ? 32767u : 65535u)) return NULL;
	const int SH = BYDIAG ? 15 : 16;             /* index bits below the key */
	uint32_t np = 64; while(np < n) np <<= 1;
	uint64_t *w;
	if(lds && np <= lds_u64) w = lds;
	else {
		uint64_t a = 0;
		if(lane == 0) a = (uint64_t)(uintptr_t)wtz_pool_alloc(pool, (size_t)np * 8);
		a = wtz_coop_bcast64(a);
		w = (uint64_t*)(uintptr_t)a;
		if(w == NULL){ *pool_bad = 1; return NULL; }
	}
	return (w == lds) ? wtz_sort_hits_body<BYDIAG, true>(hits, n, pool, lds, lds_u64, w, np, pool_bad) : wtz_sort_hits_body<BYDIAG, false>(hits, n, pool, lds, lds_u64, w, np, pool_bad);
}
#endif

/* scratch of one (pair,strand) window scan: all sized by the number of matches of the pair.  `lds` (may be NULL) is the
 * wave's LDS slice: the small order-sensitive sorts / the quick-select run there when they fit - they are chains of
 * dependent loads on a single lane, so it is the access latency (LDS ~64 clk vs L2 ~500 clk) that matters */
/* `big`: workspace in the pool for the scans whose matches do not fit the LDS slice (allocated by the first such scan of the pair, grown on demand);
 * `need_big`: set by a scan of the LDS-only kernel that met such a range - the pair is finished by the launch that carries the pool-workspace body */
/* `wf`, `wd` (n + 2 words each): the two per-match tables of the wave-parallel window merge (wtz_merge_prepare) */
typedef struct { uint32_t *ts; int32_t *as; uint32_t *wb, *we, *wo; uint64_t *tk; wtz_zhit_t *ztmp; uint64_t *lds; uint32_t lds_u64; mutable uint64_t *big; mutable uint32_t big_u64; mutable uint32_t need_big; uint32_t *wf, *wd; } wtz_winscratch_t;
struct wtz_gt_hi32 { WTZ_HDM bool operator()(uint64_t a, uint64_t b) const { return (uint32_t)(a >> 32) > (uint32_t)(b >> 32); } };

/*
 * Wave-cooperative form of the two functions above (same results).  What bounds a pair here is not arithmetic but the
 * latency of dependent loads on a single lane, so:
 *   - the merge loop reads the (off1,off2)-ordered matches through two 64-entry register windows (one hit per lane,
 *     fetched with one coalesced load; the uniform cursor reads them with v_readlane);
 *   - a window scan compacts its strand-filtered matches with ballot ranks, orders them by off2 with the wave-wide
 *     bitonic network (any off2 tie makes lane 0 redo the swap-exact sort from the original order - tie order is
 *     observable), and gathers them ONCE into LDS; the sequential sweep / median / anchor ordering of lane 0 then run
 *     out of LDS and registers only.
 * Vectors (`wins`, `anchors`) are owned by lane 0; counts and the window extent the merge loop needs are broadcast.
 * A scan whose matches do not fit the LDS slice (np + 2n + 2 > lds_u64 words) falls back to the scalar body on lane 0.
 */
typedef struct { uint32_t base, o1, o2, ll; } wtz_hitcur_t;        /* lane l holds match base + l */
WTZ_HD void wtz_hitcur_get(wtz_hitcur_t &c, const wtz_zhit_t *rs, uint32_t lim, uint32_t i, uint32_t &o1, uint32_t &o2, uint32_t &ll){
	if(i - c.base >= WTZ_NLANES){           /* uniform; also true for i < base */
		c.base = i;
		const uint32_t idx = i + WTZ_LANE;
		if(idx <= lim){ const wtz_zhit_t h = rs[idx]; c.o1 = h.o1; c.o2 = h.o2; c.ll = h.ll; } else { c.o1 = c.o2 = c.ll = 0; }
	}
	const uint32_t l = i - c.base;
	o1 = wtz_coop_lane32(c.o1, l); o2 = wtz_coop_lane32(c.o2, l); ll = wtz_coop_lane32(c.ll, l);
}

/* the part of a scan that only the scans past the early exits reach (one in seven at configs[2]): ordering, sweep, windows.  WTZ_SCAN_REST_FN / WTZ_SCAN_FN choose
 * whether it / the whole scan is inlined into the merge loop (WTZ_HD) or a function of its own (WTZ_HDN: the merge loop's cursors do not share the register file with this body) */
#ifndef WTZ_SCAN_REST_FN
#define WTZ_SCAN_REST_FN WTZ_HD
#endif
#ifndef WTZ_SCAN_FN
#define WTZ_SCAN_FN WTZ_HD
#endif
template<bool K_LDS>
WTZ_SCAN_REST_FN uint32_t wtz_scan_windows_rest(const wtz_zhit_t *rs, uint32_t dir, uint32_t beg, uint32_t end, int32_t bound,
		wtz_vec<wtz_win_t> &wins, wtz_vec<wtz_zhit_t> &anchors, const wtz_winscratch_t &sc, uint32_t zsize, uint32_t kwin, uint32_t zovl, int32_t *max_e0, uint32_t n, unsigned long long pw0, uint64_t *KB){
	const uint32_t lane = WTZ_LANE;
	/* K = the LDS slice (K_LDS: the address space stays static, DS instructions), or the pool workspace of a scan that does not fit it: the same code, the
	 * orderings then go through the LDS slice as a window */
	uint64_t *K = K_LDS ? sc.lds : KB;
	auto sort_keys = [&](uint64_t *w, uint32_t npw){
		if(K_LDS || sc.lds == NULL){ wtz_coop_sort_u64(w, npw); return; }
		uint32_t ln = 128; while(ln * 2 <= sc.lds_u64) ln <<= 1;
		wtz_coop_sort_u64_windowed(w, npw, sc.lds_u64 >= 128u ? sc.lds : NULL, ln);
	};
	uint32_t ret = 0; int32_t me0 = -0x7FFFFFFF;
	(void)pw0; (void)zsize;
	WTZ_PROF_ADD(16, pw0); WTZ_PROF_CNT(21, 1); WTZ_PROF_CNT(22, n);
	const unsigned long long pw1 = WTZ_PROF_T(); (void)pw1;
	uint32_t np = 64; while(np < n) np <<= 1;
	/* ---- order by off2 (hzm_aln.h:449).  Equal off2 = ONE candidate z-mer matched at several query positions (a start position holds one z-mer, and on the
	 * reverse strand distinct starts have distinct ends), so tied matches agree in (off2, len2) and the sweep below - which reads nothing else - computes the same
	 * overlaps, evicts a tie group as a whole and opens its windows at a group's first member whatever the order inside the group.  The order is observable in
	 * exactly two places: (a) a window whose last member is a non-last member of a tie group (the rest of the group stays outside), (b) the swap-exact ordering
	 * of a window's members by off1 when THAT has ties too (its input permutation changes).  So the first attempt takes the wave's order (ties by position in the
	 * match list), watches for (a) and (b), and only then the scan is redone with the reference's swap sequence replayed on lane 0 - round 4's profile: 1.9 M of
	 * 7.6 M scans had ties and their replays were two thirds of the ordering time. ---- */
	bool had_tie = false, exact = false;
	const uint32_t wins_n0 = wins.n, anchors_n0 = anchors.n;      /* lane 0's values are the ones used */
retry_scan:
	if(exact){
		uint32_t m = 0;
		for(uint32_t b0 = beg; b0 < end; b0 += WTZ_NLANES){      /* K in the ORIGINAL order again */
			const uint32_t idx = b0 + lane;
			bool keep = false; uint32_t o2 = 0;
			if(idx < end){ const uint32_t o1 = rs[idx].o1; o2 = rs[idx].o2; keep = (((o1 ^ o2) >> 31) == dir) && ((int32_t)(o1 & 0x7FFFFFFFu) >= bound); }
			uint32_t tot; const uint32_t pos = wtz_coop_rank(keep, &tot);
			if(keep) K[m + pos] = ((uint64_t)(o2 & 0x7FFFFFFFu) << 32) | idx;
			m += tot;
		}
		WTZ_WAVE_SYNC();
		{ const unsigned long long px = WTZ_PROF_T(); (void)px; if(lane == 0) wtz_sort_exact(K, (size_t)n, wtz_gt_hi32()); WTZ_PROF_ADD(25, px); WTZ_PROF_CNT(24, 1000); }
		WTZ_WAVE_SYNC();
	} else {
#if defined(__HIP_DEVICE_COMPILE__)
	bool in_regs = false;
	if(n <= 64u){
		/* one key per lane: the order comes out of registers */
		WTZ_WAVE_SYNC();
		const uint64_t v = wtz_wave_sort64(lane < n ? K[lane] : ~0ull);
		const uint32_t nhi = (uint32_t)__shfl_down((int)(uint32_t)(v >> 32), 1, 64);
		const bool tie = lane + 1 < n && (uint32_t)(v >> 32) == nhi;
		uint32_t any; (void)wtz_coop_rank(tie, &any);
		had_tie = any != 0;
		if(lane < n) K[lane] = v;
		WTZ_WAVE_SYNC();
		in_regs = true;
	}
	if(!in_regs)
#endif
	{
	for(uint32_t i = n + lane; i < np; i += WTZ_NLANES) K[i] = ~0ull;
	WTZ_WAVE_SYNC();
	sort_keys(K, np);
	bool tie = false;
	for(uint32_t i = lane; i + 1 < n; i += WTZ_NLANES) if((K[i] >> 32) == (K[i + 1] >> 32)) tie = true;
	uint32_t any; (void)wtz_coop_rank(tie, &any);
	had_tie = any != 0;
	}
	if(had_tie) WTZ_PROF_CNT(26, 1000);
	}
	WTZ_PROF_ADD(17, pw1);
	const unsigned long long pw2 = WTZ_PROF_T(); (void)pw2;
	wtz_zhit_t *S = (wtz_zhit_t*)(K + np);
	for(uint32_t x = lane; x < n; x += WTZ_NLANES) S[x] = rs[(uint32_t)K[x]];
	WTZ_WAVE_SYNC();
	WTZ_PROF_ADD(18, pw2);
	const unsigned long long pw3 = WTZ_PROF_T(); (void)pw3;
	unsigned long long pw4 = 0; (void)pw4;
	uint32_t n2_all = 0;
	{
		/* ---- the sweep of hzm_aln.h:451-483, one match per lane.  The running overlap is a difference of two prefix sums (all of it modulo 2^32, like the loop):
		 * entering match i adds add_i = len2_i, or end_i - end_(i-1) when it starts inside its predecessor; evicting match k takes evict_k = len2_k - (its overlap
		 * with match k + 1); the eviction cursor after match i is the first j with end_i <= off2_j + kwin (off2 ascends: a binary search), never behind the cursor of
		 * the match before (a running maximum).  So ol_i = A_i - E_j(i) for every i at once, and only the window bookkeeping - which merges an opening into the last
		 * window or starts a new one depending on the last window's best member - walks the openings in order, on uniform values read from the lanes' registers. ---- */
		uint32_t *EP = (uint32_t*)K;                    /* EP[j] = sum of evict_k over k < j; the off2 keys in K are dead after the gather */
		uint32_t accA = 0, accE = 0, jrun = 0, n2 = 0;
		uint32_t lwo = 0, lwb_o2 = 0, lwe_o2 = 0;      /* the open window: overlap and off2 of its two ends */
		for(uint32_t c0 = 0; c0 < n; c0 += WTZ_NLANES){
			const uint32_t i = c0 + lane; const bool in = i < n;
			uint32_t o2 = 0, l2 = 0, e2 = 0, add = 0, ev = 0;
			if(in){
				const wtz_zhit_t p = S[i]; o2 = ZH_OFF2(p); l2 = ZH_LEN2(p); e2 = o2 + l2;
				uint32_t lst = 0; if(i){ const wtz_zhit_t q = S[i - 1]; lst = ZH_OFF2(q) + ZH_LEN2(q); }
				add = (o2 > lst) ? l2 : e2 - lst;
				if(i + 1 < n){ const uint32_t s1 = ZH_OFF2(S[i + 1]); ev = l2 - (s1 < e2 ? e2 - s1 : 0u); }
			}
			uint32_t totA, totE;
			const uint32_t exA = wtz_coop_excl_scan(add, &totA), exE = wtz_coop_excl_scan(ev, &totE);
			if(in) EP[i] = accE + exE;
			WTZ_WAVE_SYNC();
			uint32_t jm = 0;
			if(in){ uint32_t lo = 0, hi = i; while(lo < hi){ const uint32_t mid = (lo + hi) >> 1; if(e2 > ZH_OFF2(S[mid]) + kwin) lo = mid + 1; else hi = mid; } jm = lo; }
			jm = wtz_coop_incl_max32(jm); jm = jm > jrun ? jm : jrun;
			uint32_t ol = 0, jo2 = 0;
			if(in){ ol = accA + exA + add - EP[jm]; jo2 = ZH_OFF2(S[jm]); }
			const uint32_t lastl = (n - c0 < WTZ_NLANES ? n - c0 : WTZ_NLANES) - 1;
			accA += totA; accE += totE; jrun = wtz_coop_lane32(jm, lastl);
			unsigned long long fl = wtz_coop_ballot(in && ol >= zovl);
			while(fl){
				const uint32_t l = (uint32_t)__builtin_ctzll(fl); fl &= fl - 1;
				const uint32_t ol_ = wtz_coop_lane32(ol, l), o2_ = wtz_coop_lane32(o2, l), jo2_ = wtz_coop_lane32(jo2, l), j_ = wtz_coop_lane32(jm, l);
				if(n2 && (o2_ <= lwe_o2 + kwin / 3 || jo2_ <= lwb_o2 + kwin / 3)){
					if(ol_ > lwo){ if(lane == 0){ sc.wb[n2-1] = j_; sc.we[n2-1] = c0 + l; } lwo = ol_; lwb_o2 = jo2_; lwe_o2 = o2_; }
				} else { if(lane == 0){ sc.wb[n2] = j_; sc.we[n2] = c0 + l; } lwo = ol_; lwb_o2 = jo2_; lwe_o2 = o2_; n2++; }
			}
			WTZ_WAVE_SYNC();                                 /* EP of this chunk is read before the next chunk writes beside it */
		}
#ifdef WTZ_EXP_CNT_EMPTY
		WTZ_PROF_ADD(19, pw3); pw4 = WTZ_PROF_T(); WTZ_PROF_CNT(23, n2 == 0 ? 1 : 0);      /* diagnostic build: scans whose sweep finds no window */
#else
		WTZ_PROF_ADD(19, pw3); pw4 = WTZ_PROF_T(); WTZ_PROF_CNT(23, n2);
#endif
		n2_all = n2;
		if(had_tie && !exact){          /* (a): does a window end inside a tie group? */
			uint32_t cut = 0;
			if(lane == 0) for(uint32_t wq = 0; wq < n2; wq++){ const uint32_t we = sc.we[wq]; if(we + 1 < n && ZH_OFF2(S[we + 1]) == ZH_OFF2(S[we])) cut = 1; }
			if(wtz_coop_bcast32(cut)) n2_all = 0xFFFFFFFFu;
		}
	}
	/* ---- the windows (hzm_aln.h:484-575), every step on the whole wave (round 3 walked them on lane 0: chains of dependent LDS reads).
	 *  - median diagonal (calculate_median_value returns the element of rank size / 2): the rank from the wave's sorting network (<= 64 members) or the LDS network;
	 *  - members within KWIN_MAX_OFFSET_DEV of it, compacted in S order with ballot ranks;
	 *  - their order by off1 (hzm_aln.h:519): distinct keys have ONE ascending order; equal off1 (one query z-mer matched at two candidate positions) makes the
	 *    reference's swap sequence observable and lane 0 replays it from the S order;
	 *  - anchors copied one per lane; the covered length is a sum of per-member terms that depend on the member before only (len1, or end - previous end), the box
	 *    is four min / max reductions. ---- */
	if(n2_all == 0xFFFFFFFFu){ WTZ_PROF_CNT(28, 1000); exact = true; goto retry_scan; }
	int32_t last_end1 = 0; uint32_t last_ovl = 0;
	for(uint32_t wi = 0; wi < n2_all; wi++){
		uint32_t size = 0, cnt = 0, wb = 0, we = 0;
		uint64_t *ak = K;                         /* (off1<<32 | position in S) of the members */
		if(lane == 0){ size = anchors.n; wb = sc.wb[wi]; we = sc.we[wi]; }
		size = wtz_coop_bcast32(size); wb = wtz_coop_bcast32(wb); we = wtz_coop_bcast32(we);
		const uint32_t m = we - wb + 1;
		int32_t offset;
		WTZ_WAVE_SYNC();
#if defined(__HIP_DEVICE_COMPILE__)
		if(m <= 64u){
			uint32_t key = 0xFFFFFFFFu;
			if(lane < m){ const wtz_zhit_t p = S[wb + lane]; key = (uint32_t)((int32_t)ZH_OFF1(p) - (int32_t)ZH_OFF2(p)) ^ 0x80000000u; }
			key = wtz_wave_sort32(key);
			offset = (int32_t)(wtz_coop_lane32(key, m / 2) ^ 0x80000000u);
		} else
#endif
		{
			uint32_t mp = 64; while(mp < m) mp <<= 1;                /* m <= n <= np */
			for(uint32_t x = lane; x < mp; x += WTZ_NLANES){
				uint64_t v = ~0ull;
				if(x < m){ const wtz_zhit_t p = S[wb + x]; v = (uint64_t)((uint32_t)((int32_t)ZH_OFF1(p) - (int32_t)ZH_OFF2(p)) ^ 0x80000000u); }
				ak[x] = v;
			}
			WTZ_WAVE_SYNC();
			sort_keys(ak, mp);
			offset = (int32_t)((uint32_t)ak[m / 2] ^ 0x80000000u);
			WTZ_WAVE_SYNC();
		}
		auto fill_members = [&]() -> uint32_t {   /* members within the deviation, in S order */
			uint32_t c2 = 0;
			for(uint32_t c0 = 0; c0 < m; c0 += WTZ_NLANES){
				const uint32_t j = wb + c0 + lane;
				bool keep = false; uint32_t o1 = 0;
				if(c0 + lane < m){ const wtz_zhit_t p = S[j]; const int32_t off = (int32_t)ZH_OFF1(p) - (int32_t)ZH_OFF2(p); o1 = ZH_OFF1(p); keep = !(off < offset - WTZ_KWIN_MAX_OFFSET_DEV || off > offset + WTZ_KWIN_MAX_OFFSET_DEV); }
				uint32_t tot; const uint32_t pos = wtz_coop_rank(keep, &tot);
				if(keep) ak[c2 + pos] = ((uint64_t)o1 << 32) | j;
				c2 += tot;
			}
			WTZ_WAVE_SYNC();
			return c2;
		};
		cnt = fill_members();
		if(cnt == 0) continue;
		{
			uint32_t any = 0;
#if defined(__HIP_DEVICE_COMPILE__)
			if(cnt <= 64u){
				const uint64_t v = wtz_wave_sort64(lane < cnt ? ak[lane] : ~0ull);
				const uint32_t nhi = (uint32_t)__shfl_down((int)(uint32_t)(v >> 32), 1, 64);
				const bool tie = lane + 1 < cnt && (uint32_t)(v >> 32) == nhi;
				(void)wtz_coop_rank(tie, &any);
				if(!any && lane < cnt) ak[lane] = v;
				WTZ_WAVE_SYNC();
			} else
#endif
			{
				uint32_t np2 = 64; while(np2 < cnt) np2 <<= 1;            /* cnt <= n <= np: the padding stays in front of S */
				for(uint32_t i = cnt + lane; i < np2; i += WTZ_NLANES) ak[i] = ~0ull;
				WTZ_WAVE_SYNC();
				sort_keys(ak, np2);
				bool tie = false;
				for(uint32_t i = lane; i + 1 < cnt; i += WTZ_NLANES) if((ak[i] >> 32) == (ak[i + 1] >> 32)) tie = true;
				(void)wtz_coop_rank(tie, &any);
				WTZ_PROF_CNT(29, 1000);
				if(any){ WTZ_WAVE_SYNC(); (void)fill_members(); }      /* back to the S order for the exact path */
			}
			if(any){
				if(had_tie && !exact){         /* (b): both orders have ties - the scan again, with the reference's swap sequence for off2 */
					if(lane == 0){ wins.n = wins_n0; anchors.n = anchors_n0; }
					ret = 0; me0 = -0x7FFFFFFF; exact = true;
					WTZ_PROF_CNT(30, 1000);
					goto retry_scan;
				}
				const unsigned long long px = WTZ_PROF_T(); (void)px;
				if(lane == 0) wtz_sort_exact(ak, (size_t)cnt, wtz_gt_hi32());
				WTZ_PROF_ADD(27, px); WTZ_PROF_CNT(31, 1000);
				WTZ_WAVE_SYNC();
			}
		}
		uint32_t stop = 0; uint64_t abase = 0;
		if(lane == 0){ if(!anchors.reserve(size + cnt)) stop = 1; else abase = (uint64_t)(uintptr_t)anchors.a; }
		if(wtz_coop_bcast32(stop)) break;
		abase = wtz_coop_bcast64(abase);
		wtz_zhit_t *A = (wtz_zhit_t*)(uintptr_t)abase + size;
		uint32_t ol = 0, b0 = 0x7FFFFFFFu, e0 = 0, b1 = 0x7FFFFFFFu, e1 = 0;
		for(uint32_t c0 = 0; c0 < cnt; c0 += WTZ_NLANES){
			const uint32_t k = c0 + lane; uint32_t add = 0;
			if(k < cnt){
				const wtz_zhit_t p = S[(uint32_t)ak[k]];
				A[k] = p;
				uint32_t lst = 0; if(k){ const wtz_zhit_t q = S[(uint32_t)ak[k - 1]]; lst = ZH_OFF1(q) + ZH_LEN1(q); }
				const uint32_t o1 = ZH_OFF1(p), x1 = o1 + ZH_LEN1(p), o2 = ZH_OFF2(p), x2 = o2 + ZH_LEN2(p);
				add = (o1 > lst) ? ZH_LEN1(p) : x1 - lst;
				b0 = o1 < b0 ? o1 : b0; e0 = x1 > e0 ? x1 : e0; b1 = o2 < b1 ? o2 : b1; e1 = x2 > e1 ? x2 : e1;
			}
			uint32_t tot; (void)wtz_coop_excl_scan(add, &tot); ol += tot;
		}
		b0 = wtz_coop_min32(b0); b1 = wtz_coop_min32(b1); e0 = ~wtz_coop_min32(~e0); e1 = ~wtz_coop_min32(~e1);
		if(ol * 2 < zovl) continue;
		if(ret && ((int32_t)e1 <= (int32_t)((uint32_t)last_end1 + kwin / 3) && ol <= last_ovl)) continue;
		if(lane == 0){
			wtz_win_t w;
			w.pb2 = 0; w.closed = 0; w.dir = (uint8_t)dir; w.pad = 0;
			w.beg[0] = (int32_t)b0; w.beg[1] = (int32_t)b1; w.end[0] = (int32_t)e0; w.end[1] = (int32_t)e1;
			anchors.n = size + cnt;
			w.anchors[0] = size; w.anchors[1] = anchors.n;
			w.ovl = WTZ_OVL29(ol);
			if(!wins.push(w)) stop = 1;                /* pool exhausted: count only what is in the vector */
		}
		if(wtz_coop_bcast32(stop)) break;
		ret++;
		last_end1 = (int32_t)e1; last_ovl = WTZ_OVL29(ol);
		if(me0 < (int32_t)e0) me0 = (int32_t)e0;
		WTZ_WAVE_SYNC();
	}
	WTZ_PROF_ADD(20, pw4);
	WTZ_WAVE_SYNC();
	*max_e0 = (int32_t)wtz_coop_bcast32((uint32_t)me0);
	return wtz_coop_bcast32(ret);
}

/* the pool-workspace instance as a function of its own: it is reached by a handful of pairs per launch, and inlined it doubled the scan's code inside the merge
 * loop (218 spilled registers, K_pair 777 -> 1 428 ms) */
WTZ_DN uint32_t wtz_scan_windows_rest_pool(const wtz_zhit_t *rs, uint32_t dir, uint32_t beg, uint32_t end, int32_t bound,
		wtz_vec<wtz_win_t> &wins, wtz_vec<wtz_zhit_t> &anchors, const wtz_winscratch_t &sc, uint32_t zsize, uint32_t kwin, uint32_t zovl, int32_t *max_e0, uint32_t n, unsigned long long pw0, uint64_t *KB){
	return wtz_scan_windows_rest<false>(rs, dir, beg, end, bound, wins, anchors, sc, zsize, kwin, zovl, max_e0, n, pw0, KB);
}

template<bool ZBIG>
WTZ_SCAN_FN uint32_t wtz_scan_windows_coop(const wtz_zhit_t *rs, uint32_t dir, uint32_t beg, uint32_t end, int32_t bound,
		wtz_vec<wtz_win_t> &wins, wtz_vec<wtz_zhit_t> &anchors, const wtz_winscratch_t &sc, uint32_t zsize, uint32_t kwin, uint32_t zovl, int32_t *max_e0){
	const uint32_t lane = WTZ_LANE;
	uint64_t *K = sc.lds;
	uint32_t n = 0;
	beg = wtz_coop_bcast32(beg); end = wtz_coop_bcast32(end); bound = (int32_t)wtz_coop_bcast32((uint32_t)bound); dir = wtz_coop_bcast32(dir);      /* uniform by construction: scalar loop bounds */
	zsize = wtz_coop_bcast32(zsize); kwin = wtz_coop_bcast32(kwin); zovl = wtz_coop_bcast32(zovl);
	const unsigned long long pw0 = WTZ_PROF_T(); (void)pw0;
	/* ---- strand / bound filter (the prefix skip of hzm_aln.h:425-431 is the same predicate: off1 is non-decreasing) ---- */
	constexpr uint32_t RC = 4;                         /* a range of up to RC matches per lane is read ONCE: count, early-exit bound and key fill run on the off2 values kept in registers */
#ifndef WTZ_NO_SCAN_CACHE
	/* ... and its ordering workspace (power of two >= 64 keys + two words per match + 2) fits the slice whatever the count turns out to be */
	const bool cached = (K != NULL) && (end - beg <= WTZ_NLANES * RC) && (sc.lds_u64 >= (WTZ_NLANES * RC > 64u ? WTZ_NLANES * RC : 64u) + 2u * WTZ_NLANES * RC + 2u);
#else
	const bool cached = false;
#endif
	if(cached){
		uint32_t q2[RC]; uint32_t kpm = 0;              /* off2 of this lane's matches; bit c: match c passes the strand / bound filter */
		#pragma unroll
		for(uint32_t c = 0; c < RC; c++){
			const uint32_t idx = beg + c * WTZ_NLANES + lane;
			q2[c] = 0;
			if(idx < end){ const uint32_t o1 = rs[idx].o1, o2 = rs[idx].o2; q2[c] = o2 & 0x7FFFFFFFu; if((((o1 ^ o2) >> 31) == dir) && ((int32_t)(o1 & 0x7FFFFFFFu) >= bound)) kpm |= 1u << c; }
		}
		n = 0;
		#pragma unroll
		for(uint32_t c = 0; c < RC; c++) n += (uint32_t)__builtin_popcountll(wtz_coop_ballot((kpm >> c) & 1u));
		if(n * zsize < zovl){ WTZ_PROF_ADD(12, pw0); WTZ_PROF_CNT(7, 1); return 0; }
#ifndef WTZ_NO_SCAN_PRECHECK
		{       /* the exact early exit described below, on the cached matches (their len2 is read here, by the scans that get this far) */
			uint32_t mn = 0xFFFFFFFFu, mx = 0;
			#pragma unroll
			for(uint32_t c = 0; c < RC; c++) if((kpm >> c) & 1u){ mn = q2[c] < mn ? q2[c] : mn; mx = q2[c] > mx ? q2[c] : mx; }
			mn = wtz_coop_min32(mn); mx = ~wtz_coop_min32(~mx);
			const uint32_t bw = kwin >= 8 ? kwin / 8 : 1;
			const uint32_t nadj = (kwin - 1) / bw + 2;      /* the bins are aligned to mn, not to the window: kwin - 1 columns starting anywhere inside a bin reach into one more */
			const uint32_t nb = (mx - mn) / bw + 1 + nadj;
			if(nb <= sc.lds_u64 * 2){
				uint32_t *B = (uint32_t*)K;
				for(uint32_t b = lane; b < nb; b += WTZ_NLANES) B[b] = 0;
				WTZ_WAVE_SYNC();
				#pragma unroll
				for(uint32_t c = 0; c < RC; c++) if((kpm >> c) & 1u){
					const uint32_t bin = (q2[c] - mn) / bw, l2 = rs[beg + c * WTZ_NLANES + lane].ll >> 16;
#if defined(__HIP_DEVICE_COMPILE__)
					atomicAdd(&B[bin], l2);
#else
					B[bin] += l2;
#endif
				}
				WTZ_WAVE_SYNC();
				uint32_t best = 0;
				for(uint32_t b = lane; b + nadj <= nb; b += WTZ_NLANES){ uint32_t v = 0; for(uint32_t t = 0; t < nadj; t++) v += B[b + t]; best = v > best ? v : best; }
				best = ~wtz_coop_min32(~best);
				WTZ_WAVE_SYNC();
				if(best < zovl){ WTZ_PROF_CNT(23, 0); WTZ_PROF_ADD(12, pw0); WTZ_PROF_CNT(7, 1); return 0; }
			}
		}
#endif
		uint32_t m = 0;
		#pragma unroll
		for(uint32_t c = 0; c < RC; c++){
			const bool kp = (kpm >> c) & 1u;
			uint32_t tot; const uint32_t pos = wtz_coop_rank(kp, &tot);
			if(kp) K[m + pos] = ((uint64_t)q2[c] << 32) | (beg + c * WTZ_NLANES + lane);
			m += tot;
		}
	}
	for(int pass = 0; pass < 2 && !cached; pass++){
		n = 0;
		for(uint32_t b0 = beg; b0 < end; b0 += WTZ_NLANES){
			const uint32_t idx = b0 + lane;
			bool keep = false; uint32_t o2 = 0;
			if(idx < end){ const uint32_t o1 = rs[idx].o1; o2 = rs[idx].o2; keep = (((o1 ^ o2) >> 31) == dir) && ((int32_t)(o1 & 0x7FFFFFFFu) >= bound); }
			uint32_t tot; const uint32_t pos = wtz_coop_rank(keep, &tot);
			if(pass && keep) K[n + pos] = ((uint64_t)(o2 & 0x7FFFFFFFu) << 32) | idx;
			n += tot;
		}
		if(pass) break;
		if(n * zsize < zovl){ WTZ_PROF_ADD(12, pw0); WTZ_PROF_CNT(7, 1); return 0; }      /* profiler: 12 = time in scans that end before the ordering, 7 = how many */
#ifndef WTZ_NO_SCAN_PRECHECK
		/* ---- exact early exit (round 3): can the sweep of hzm_aln.h:451-483 reach `ol >= zovl` at all?  In off2 order the running overlap is
		 * ol = sum of len2 over the in-window matches j..i minus the overlaps of neighbours (a match adds len2 minus its overlap with the one
		 * before, an eviction takes len2 minus the overlap with the one behind; ends are monotone in off2 for the matches of one strand), so
		 * ol <= sum of len2 over j..i, and the while loop keeps off2[i] - off2[j] < kwin: the in-window matches lie in nadj adjacent bins of
		 * kwin / 8 columns (nine bins = 1.125 kwin: eight would miss a window that starts late in its first bin).  If no such run of bins holds zovl bases of matches, no window can open, n2 stays 0 and the scan returns 0 - without
		 * the off2 ordering, the gather and the lane-0 sweep.  Tie order cannot matter: the bound is over sets. ---- */
		if(K != NULL){
			uint32_t mn = 0xFFFFFFFFu, mx = 0;
			for(uint32_t b0 = beg; b0 < end; b0 += WTZ_NLANES){
				const uint32_t idx = b0 + lane;
				if(idx < end){ const uint32_t o1 = rs[idx].o1, o2 = rs[idx].o2; if((((o1 ^ o2) >> 31) == dir) && ((int32_t)(o1 & 0x7FFFFFFFu) >= bound)){ const uint32_t f = o2 & 0x7FFFFFFFu; mn = f < mn ? f : mn; mx = f > mx ? f : mx; } }
			}
			mn = wtz_coop_min32(mn); mx = ~wtz_coop_min32(~mx);
			const uint32_t bw = kwin >= 8 ? kwin / 8 : 1;          /* bins of an eighth of the window: the in-window matches span < kwin columns = at most nadj adjacent bins */
			const uint32_t nadj = (kwin - 1) / bw + 2;      /* the bins are aligned to mn, not to the window: kwin - 1 columns starting anywhere inside a bin reach into one more */
			const uint32_t nb = (mx - mn) / bw + 1 + nadj;         /* nadj empty bins behind the last: every run of nadj bins starting at a used bin is inside the array */
			if(nb <= sc.lds_u64 * 2){
				uint32_t *B = (uint32_t*)K;
				for(uint32_t b = lane; b < nb; b += WTZ_NLANES) B[b] = 0;
				WTZ_WAVE_SYNC();
				for(uint32_t b0 = beg; b0 < end; b0 += WTZ_NLANES){
					const uint32_t idx = b0 + lane;
					if(idx < end){
						const uint32_t o1 = rs[idx].o1, o2 = rs[idx].o2;
						if((((o1 ^ o2) >> 31) == dir) && ((int32_t)(o1 & 0x7FFFFFFFu) >= bound)){
							const uint32_t bin = ((o2 & 0x7FFFFFFFu) - mn) / bw, l2 = rs[idx].ll >> 16;
#if defined(__HIP_DEVICE_COMPILE__)
							atomicAdd(&B[bin], l2);
#else
							B[bin] += l2;
#endif
						}
					}
				}
				WTZ_WAVE_SYNC();
				uint32_t best = 0;
				for(uint32_t b = lane; b + nadj <= nb; b += WTZ_NLANES){ uint32_t v = 0; for(uint32_t t = 0; t < nadj; t++) v += B[b + t]; best = v > best ? v : best; }
				best = ~wtz_coop_min32(~best);
				WTZ_WAVE_SYNC();                                  /* K is written again by the second pass */
				if(best < zovl){ WTZ_PROF_CNT(23, 0); WTZ_PROF_ADD(12, pw0); WTZ_PROF_CNT(7, 1); return 0; }
			}
		}
#endif
		uint32_t np0 = 64; while(np0 < n) np0 <<= 1;
		if(K == NULL || np0 + 2 * n + 2 > sc.lds_u64){
			/* does not fit the LDS slice (a few hundred matches of one strand inside one window: repeats): the same wave-parallel scan on a workspace in the pool.
			 * (Round 3 ran these on lane 0 - a sort of thousands of matches by dependent memory accesses; the heaviest pair of a range took half of its launch.) */
			if(!ZBIG){ sc.need_big = 1; *max_e0 = -0x7FFFFFFF; return 0; }
			const uint32_t need = np0 + 2 * n + 2;
#ifndef WTZ_PAIR_NO_POOL_SCAN
			if(sc.big_u64 < need){
				const uint32_t cap = need > 2 * sc.big_u64 ? need : 2 * sc.big_u64;
				uint64_t pa = 0;
				if(lane == 0) pa = (uint64_t)(uintptr_t)wtz_pool_alloc(wins.pool, (size_t)cap * 8 + 16);
				pa = wtz_coop_bcast64(pa);
				if(pa){ sc.big = (uint64_t*)(uintptr_t)pa; sc.big_u64 = cap; }
			}
#endif
#ifdef WTZ_PAIR_NO_POOL_SCAN
			if(false){
#else
			if(sc.big_u64 >= need){
#endif
				uint64_t *KB = sc.big; uint32_t m = 0;
				for(uint32_t b0 = beg; b0 < end; b0 += WTZ_NLANES){
					const uint32_t idx = b0 + lane;
					bool keep = false; uint32_t o2 = 0;
					if(idx < end){ const uint32_t o1 = rs[idx].o1; o2 = rs[idx].o2; keep = (((o1 ^ o2) >> 31) == dir) && ((int32_t)(o1 & 0x7FFFFFFFu) >= bound); }
					uint32_t tot; const uint32_t pos = wtz_coop_rank(keep, &tot);
					if(keep) KB[m + pos] = ((uint64_t)(o2 & 0x7FFFFFFFu) << 32) | idx;
					m += tot;
				}
				if constexpr(ZBIG){       /* a real call: its results come back in vector registers - say again that they are uniform */
					const uint32_t r = wtz_scan_windows_rest_pool(rs, dir, beg, end, bound, wins, anchors, sc, zsize, kwin, zovl, max_e0, n, pw0, KB);
					*max_e0 = (int32_t)wtz_coop_bcast32((uint32_t)*max_e0);
					return wtz_coop_bcast32(r);
				}
			} else {                                      /* no room in the pool either: wtz_pool_alloc has flagged the pool, the stage ends in WTZ_E_POOL and the range is redone in halves */
				if(lane == 0) wins.bad = 1;
				*max_e0 = -0x7FFFFFFF;
				return 0;
			}
		}
	}
	return wtz_scan_windows_rest<true>(rs, dir, beg, end, bound, wins, anchors, sc, zsize, kwin, zovl, max_e0, n, pw0, NULL);
}

template<bool ZBIG>
WTZ_HD uint32_t wtz_merge_windows_coop(const wtz_zhit_t *rs, uint32_t n_rs, uint32_t dir, wtz_vec<wtz_win_t> &wins, wtz_vec<wtz_zhit_t> &anchors,
		const wtz_winscratch_t &sc, uint32_t zsize, uint32_t kwin, uint32_t kstep, uint32_t zovl){
	/* every value this loop branches on is the same in all lanes - say so (v_readfirstlane): without it the compiler keeps the cursors in vector registers and
	 * runs each of the ~4 000 iterations per pair through exec-mask save / restore sequences (round 4's profile: this loop, not the scans, was 55 % of K_pair) */
	n_rs = wtz_coop_bcast32(n_rs); dir = wtz_coop_bcast32(dir); zsize = wtz_coop_bcast32(zsize); kwin = wtz_coop_bcast32(kwin); kstep = wtz_coop_bcast32(kstep); zovl = wtz_coop_bcast32(zovl);
	const uint32_t P_off1 = 0x1FFFFFu, P_len1 = 0x3FFu, lim = n_rs + 1;
	uint32_t i, j, n, ol, ol2, lst, wlst, s, t, ret, o1, o2, ll;
	uint32_t p_off1, p_len1, p0_off1, p0_len1, p1_off1, p1_len1;
	int32_t nxt;
	wtz_hitcur_t ci, cj; ci.base = cj.base = 0x80000000u; ci.o1 = ci.o2 = ci.ll = cj.o1 = cj.o2 = cj.ll = 0;
	ol = 0; lst = 0; wlst = 0; ret = 0;
	for(j = 0; j < n_rs; j++){ wtz_hitcur_get(cj, rs, lim, j, o1, o2, ll); if(((o1 ^ o2) >> 31) ^ dir) continue; break; }
	if(j == n_rs) return 0;
	wtz_hitcur_get(cj, rs, lim, j, o1, o2, ll);
	p0_off1 = o1 & 0x7FFFFFFFu; p0_len1 = ll & 0xFFFFu;
	p_off1 = p_len1 = 0;
	for(i = j; i <= n_rs; i++){
		if(i < n_rs){
			wtz_hitcur_get(ci, rs, lim, i, o1, o2, ll);
			if(((o1 ^ o2) >> 31) ^ dir) continue;
			p_off1 = o1 & 0x7FFFFFFFu; p_len1 = ll & 0xFFFFu;
		} else { p_off1 = P_off1; p_len1 = P_len1; }
		if(p_off1 > p0_off1 + kwin){
			if(ol >= zovl){
				int32_t me0 = 0;
				n = wtz_scan_windows_coop<ZBIG>(rs, dir, j, i, (int32_t)wlst, wins, anchors, sc, zsize, kwin, zovl, &me0);
				if(!ZBIG && sc.need_big) return 0;
				if(n){
					if((int32_t)wlst < me0 + 20) wlst = (uint32_t)(me0 + 20);
					ret += n;
					p0_off1 = p_off1; p0_len1 = p_len1; ol = p_len1; lst = p_off1 + p_len1; j = i;
				} else {
					nxt = (int32_t)(p0_off1 + kstep);
					while((int32_t)p0_off1 < nxt && j < i){
						++j; wtz_hitcur_get(cj, rs, lim, j, o1, o2, ll); p1_off1 = o1 & 0x7FFFFFFFu; p1_len1 = ll & 0xFFFFu;
						s = WTZ_MAX(p0_off1, p1_off1);
						t = WTZ_MIN(p0_off1 + p0_len1, p1_off1 + p1_len1);
						ol2 = s < t ? t - s : 0;
						ol = ol + ol2 - p0_len1;
						p0_off1 = p1_off1; p0_len1 = p1_len1;
					}
				}
			}
			if(p_off1 == P_off1) break;
			while(p_off1 > p0_off1 + kwin){
				++j; wtz_hitcur_get(cj, rs, lim, j, o1, o2, ll); p1_off1 = o1 & 0x7FFFFFFFu; p1_len1 = ll & 0xFFFFu;
				s = WTZ_MAX(p0_off1, p1_off1);
				t = WTZ_MIN(p0_off1 + p0_len1, p1_off1 + p1_len1);
				ol2 = s < t ? t - s : 0;
				ol = ol + ol2 - p0_len1;
				p0_off1 = p1_off1; p0_len1 = p1_len1;
			}
		} else {
			if(p_off1 >= lst) ol += p_len1;
			else if((int32_t)(p_off1 + p_len1) > (int32_t)lst) ol += p_off1 + p_len1 - lst;
			else continue;
			lst = p_off1 + p_len1;
		}
	}
	return ret;
}

/* ---------------------------------------------------------------------------------------------------------------------------------------------------
 * The window merge on the whole wavefront (round 5).  wtz_merge_windows_coop above is the reference's loop (hzm_aln.h:580-656) with every lane doing the same
 * scalar work: ~8 000 dependent steps per pair, ~40 scalar instructions each, through the ONE scalar unit the 16 waves of a CU share - three quarters of K_pair.
 * Between two window scans the loop is a function of tables that do not depend on its state:
 *   - the matches are in off1 order (both strands), so "match i lies beyond off1[j] + kwin" is  j < F[i]  with  F[i] = the first index whose off1 + kwin reaches
 *     off1[i]  (non-decreasing in i), and the `while` that moves the window start there is  j <- max(j, F[i]);  the start before step i is therefore the running
 *     maximum of F over the matches of the strand taken so far;
 *   - moving the start from x to y changes ol by  D[x] + ... + D[y-1],  D[k] = overlap(k, k+1) - len1[k]  (u32 arithmetic as the reference's, over ALL
 *     matches: the reference's cursor does not look at the strand, hzm_aln.h:626,640) = PD[y] - PD[x] with PD the running sum of D;
 *   - a match inside the window adds  end - max(off1, lst)  when that is positive, and lst is the running maximum of the ends of the matches that were inside
 *     the window when they came by (a match that finds the window too short is never added: hzm_aln.h:606-647).
 * So 64 consecutive matches are taken at once: running maxima and one running sum give every lane the state the loop would have in front of ITS match; the
 * first lane whose match finds the window too short with ol >= zovl (a scan is due) - or that carries the sentinel offset the reference breaks on - is handled
 * exactly like the loop does, with the state of that lane, and the sweep goes on behind it.  Scans and everything they decide are untouched.
 * F and PD are made once per pair (wtz_merge_prepare) and serve both strands.  Host emulation: the same code with one lane.
 * --------------------------------------------------------------------------------------------------------------------------------------------------- */
#if defined(__HIP_DEVICE_COMPILE__)
WTZ_D uint32_t wtz_coop_excl_max32(uint32_t v){ const uint32_t in = wtz_coop_incl_max32(v); const uint32_t up = (uint32_t)__shfl_up((int)in, 1, 64); return WTZ_LANE ? up : 0u; }
WTZ_D uint32_t wtz_coop_all_max32(uint32_t v){ return (uint32_t)__builtin_amdgcn_readlane((int)wtz_coop_incl_max32(v), 63); }
WTZ_D uint32_t wtz_coop_first_lane(unsigned long long m){ return (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_ctzll(m)); }
#else
WTZ_COOP_HOST uint32_t wtz_coop_excl_max32(uint32_t){ return 0u; }
WTZ_COOP_HOST uint32_t wtz_coop_all_max32(uint32_t v){ return v; }
WTZ_COOP_HOST uint32_t wtz_coop_first_lane(unsigned long long){ return 0u; }
#endif

/* F (sc.wf) and PD (sc.wd) of a pair's ordered matches; rs[n_rs] is the readable zero element behind them */
WTZ_HD void wtz_merge_prepare(const wtz_zhit_t *rs, uint32_t n_rs, const wtz_winscratch_t &sc, uint32_t kwin){
	const uint32_t lane = WTZ_LANE;
	n_rs = wtz_coop_bcast32(n_rs); kwin = wtz_coop_bcast32(kwin);
	uint32_t f_carry = 0, d_carry = 0;
	for(uint32_t b0 = 0; b0 < n_rs; b0 += WTZ_NLANES){
		const uint32_t i = b0 + lane; const bool in = i < n_rs;
		uint32_t a = 0, l = 0, a1 = 0, l1 = 0;
		if(in){ a = ZH_OFF1(rs[i]); l = ZH_LEN1(rs[i]); a1 = ZH_OFF1(rs[i + 1]); l1 = ZH_LEN1(rs[i + 1]); }
		/* F: the answer lies in [F of the block before, i] */
		uint32_t lo = f_carry, hi = in ? i : f_carry;
		while(wtz_coop_ballot(lo < hi) != 0ull){
			if(lo < hi){ const uint32_t mid = lo + (hi - lo) / 2u; if(ZH_OFF1(rs[mid]) + kwin >= a) hi = mid; else lo = mid + 1u; }
		}
		if(in) sc.wf[i] = lo;
		f_carry = wtz_coop_all_max32(in ? lo : 0u);
		/* PD: running sum of D in front of i */
		const uint32_t s = WTZ_MAX(a, a1), t = WTZ_MIN(a + l, a1 + l1);
		const uint32_t d = in ? ((s < t ? t - s : 0u) - l) : 0u;
		uint32_t tot; const uint32_t ex = wtz_coop_excl_scan(d, &tot);
		if(in) sc.wd[i] = d_carry + ex;
		d_carry += tot;
	}
	if(lane == 0) sc.wd[n_rs] = d_carry;
	WTZ_WAVE_SYNC();
}

template<bool ZBIG>
WTZ_HD uint32_t wtz_merge_windows_wave(const wtz_zhit_t *rs, uint32_t n_rs, uint32_t dir, wtz_vec<wtz_win_t> &wins, wtz_vec<wtz_zhit_t> &anchors,
		const wtz_winscratch_t &sc, uint32_t zsize, uint32_t kwin, uint32_t kstep, uint32_t zovl){
	n_rs = wtz_coop_bcast32(n_rs); dir = wtz_coop_bcast32(dir); zsize = wtz_coop_bcast32(zsize); kwin = wtz_coop_bcast32(kwin); kstep = wtz_coop_bcast32(kstep); zovl = wtz_coop_bcast32(zovl);
	const uint32_t SENT_OFF = 0x1FFFFFu, lane = WTZ_LANE;
	const uint32_t *F = sc.wf, *PD = sc.wd;
	uint32_t j = n_rs;                                   /* the first match of the strand */
	for(uint32_t b0 = 0; b0 < n_rs; b0 += WTZ_NLANES){
		const uint32_t i = b0 + lane;
		const unsigned long long m = wtz_coop_ballot(i < n_rs && ZH_STRAND(rs[i]) == dir);
		if(m){ j = b0 + wtz_coop_first_lane(m); break; }
	}
	if(j >= n_rs) return 0;
	uint32_t ol = 0, lst = 0, wlst = 0, ret = 0, i0 = j;
	while(i0 < n_rs){
		const uint32_t i = i0 + lane; const bool in = i < n_rs;
		uint32_t o1 = 0, o2 = 0, ll = 0, fv = 0;
		if(in){ const wtz_zhit_t h = rs[i]; o1 = h.o1; o2 = h.o2; ll = h.ll; fv = F[i]; }
		const bool act = in && (((o1 ^ o2) >> 31) == dir);
		const uint32_t a = o1 & 0x7FFFFFFFu, l = ll & 0xFFFFu, e = a + l;
		const uint32_t fc = act ? (fv > j ? fv : j) : 0u;
		const uint32_t jx = wtz_coop_excl_max32(fc), jcur = jx > j ? jx : j;               /* window start in front of this lane's step */
		const bool ov = act && fc > jcur;                                                    /* the window is too short for this match */
		const uint32_t ein = (act && !ov) ? e : 0u;
		const uint32_t lx = wtz_coop_excl_max32(ein), lstb = lx > lst ? lx : lst;            /* lst in front of this lane's step */
		uint32_t delta = 0;
		if(ov) delta = PD[fc] - PD[jcur];
		else if(act){ if(a >= lstb) delta = l; else if((int32_t)e > (int32_t)lstb) delta = e - lstb; }
		uint32_t tot; const uint32_t olb = ol + wtz_coop_excl_scan(delta, &tot);            /* ol in front of this lane's step */
		const unsigned long long stop = wtz_coop_ballot(ov && (olb >= zovl || a == SENT_OFF));
		if(stop == 0ull){
			ol += tot;
			const uint32_t lm = wtz_coop_all_max32(ein), jm = wtz_coop_all_max32(fc);
			if(lm > lst) lst = lm;
			if(jm > j) j = jm;
			i0 += WTZ_NLANES;
			continue;
		}
		/* the step of the first such lane, as the loop takes it (hzm_aln.h:606-645) */
		const uint32_t tl = wtz_coop_first_lane(stop), it = i0 + tl;
		ol = wtz_coop_lane32(olb, tl); lst = wtz_coop_lane32(lstb, tl); j = wtz_coop_lane32(jcur, tl);
		const uint32_t pa = wtz_coop_lane32(a, tl), pl = wtz_coop_lane32(l, tl);
		bool taken = false;
		if(ol >= zovl){
			int32_t me0 = 0;
			const uint32_t n = wtz_scan_windows_coop<ZBIG>(rs, dir, j, it, (int32_t)wlst, wins, anchors, sc, zsize, kwin, zovl, &me0);
			if(!ZBIG && sc.need_big) return 0;
			if(n){
				if((int32_t)wlst < me0 + 20) wlst = (uint32_t)(me0 + 20);
				ret += n;
				ol = pl; lst = pa + pl; j = it; taken = true;
			} else if(kstep){
				/* the window start moves on by kstep bases (never beyond this match): the first index from j whose off1 reaches off1[j] + kstep */
				const uint32_t nxt = (uint32_t)wtz_coop_bcast32(ZH_OFF1(rs[j])) + kstep;
				uint32_t jn = it;
				for(uint32_t b0 = j; b0 < it; b0 += WTZ_NLANES){
					const uint32_t k = b0 + lane;
					const unsigned long long m = wtz_coop_ballot(k < it && (int32_t)ZH_OFF1(rs[k]) >= (int32_t)nxt);
					if(m){ jn = b0 + wtz_coop_first_lane(m); break; }
				}
				ol += wtz_coop_bcast32(PD[jn]) - wtz_coop_bcast32(PD[j]);
				j = jn;
			}
		}
		if(pa == SENT_OFF) return ret;                                   /* hzm_aln.h:633: the sentinel value ends the loop, also on a real match */
		if(!taken){
			const uint32_t fi = wtz_coop_bcast32(F[it]);
			if(fi > j){ ol += wtz_coop_bcast32(PD[fi]) - wtz_coop_bcast32(PD[j]); j = fi; }
		}
		i0 = it + 1u;
	}
	/* the sentinel behind the last match (hzm_aln.h:589): one more scan when the window has enough in it; nothing after it is observable */
	if(SENT_OFF > wtz_coop_bcast32(ZH_OFF1(rs[j])) + kwin && ol >= zovl){
		int32_t me0 = 0;
		const uint32_t n = wtz_scan_windows_coop<ZBIG>(rs, dir, j, n_rs, (int32_t)wlst, wins, anchors, sc, zsize, kwin, zovl, &me0);
		if(!ZBIG && sc.need_big) return 0;
		ret += n;
	}
	return ret;
}

/* ---------------------------------------------------------------------------------------------------------------------------------------------------
 * The window chain of a strand on the whole wavefront (round 6; chaining_wtseedv, hzm_aln.h:658-713).
 * Window i - once its own weight is final - hands  weight_i - 0.05 * |gap0 - gap1|  on to every LATER window j that starts behind its end on both axes with a
 * diagonal shift of at most W, and the reference's loop over j STOPS at the first j that lies more than W behind i on BOTH axes (hzm_aln.h:687; windows in front
 * of that j that lie beyond W on one axis only are skipped by the band test, not by the stop).  Nothing about that needs the loop: with one window per lane the
 * stop is the first set bit of a ballot, the lanes in front of it update their own weight / predecessor, and window i's end and weight are read from its lane
 * (v_readlane: i is the scalar loop counter).  The sequential remainder is the reference's own dependence - n steps over i, a window's weight must be final
 * before it is handed on - instead of n^2 / 2 dependent loads on lane 0 (round 5: `if(lane != 0) continue; wtz_chain_windows(...)`).
 * Entered by every lane with uniform arguments.  Returns the chain's length on the query axis (uniform); chain members get closed = 0, all others 1.
 * More than 64 windows of one strand (a read covered for > 50 kb): the same steps with the per-window state in `st` (2 n words) and the windows read from the
 * pool 64 at a time; the host emulation (one lane) always takes this form.
 * --------------------------------------------------------------------------------------------------------------------------------------------------- */
WTZ_HD int32_t wtz_chain_windows_wave(wtz_win_t *wins, uint32_t n, int32_t W, int32_t *st){
	const uint32_t lane = WTZ_LANE;
	const float shift_cost = 0.05f;                            /* band_penalty, hzm_aln.h:665; the product is truncated to int (hzm_aln.h:690) */
#if defined(__HIP_DEVICE_COMPILE__)
	if(n <= 64u){
		const bool mine = lane < n;
		int32_t qb = 0, qe = 0, tb = 0, te = 0, own = 0;
		if(mine){ qb = wins[lane].beg[0]; qe = wins[lane].end[0]; tb = wins[lane].beg[1]; te = wins[lane].end[1]; own = (int32_t)wins[lane].ovl; }
		int32_t acc = 0, pred = -1;                             /* weight handed to this lane's window so far, and by whom */
		int32_t top = -1000000, top_at = -1;
		for(uint32_t i = 0; i < n; i++){
			const int32_t wi = __builtin_amdgcn_readlane(acc, (int)i) + __builtin_amdgcn_readlane(own, (int)i);
			if(wi > top){ top = wi; top_at = (int32_t)i; }
			const int32_t g0 = qb - __builtin_amdgcn_readlane(qe, (int)i), g1 = tb - __builtin_amdgcn_readlane(te, (int)i);
			const bool later = mine && lane > i;
			const unsigned long long far = __ballot(later && g0 > W && g1 > W);
			const uint32_t stop_at = far ? (uint32_t)__builtin_ctzll(far) : 64u;
			if(later && lane < stop_at && g0 >= 0 && g1 >= 0){
				const int32_t shift = g0 > g1 ? g0 - g1 : g1 - g0;
				if(shift <= W){
					const int32_t offer = wi - (int32_t)((float)shift * shift_cost);
					if(acc < offer){ acc = offer; pred = (int32_t)i; }
				}
			}
		}
		unsigned long long chain = 0ull; int32_t span = 0;
		for(int32_t k = top_at; k >= 0; k = __builtin_amdgcn_readlane(pred, k)){
			chain |= 1ull << k;
			span += __builtin_amdgcn_readlane(qe, k) - __builtin_amdgcn_readlane(qb, k);
		}
		if(mine) wins[lane].closed = ((chain >> lane) & 1ull) ? 0 : 1;
		WTZ_WAVE_SYNC();
		return span;
	}
#endif
	int32_t *acc = st, *pred = st + n;
	for(uint32_t j = lane; j < n; j += WTZ_NLANES){ acc[j] = 0; pred[j] = -1; wins[j].closed = 1; }
	WTZ_WAVE_SYNC();
	int32_t top = -1000000, top_at = -1;
	for(uint32_t i = 0; i < n; i++){
		const int32_t wi = (int32_t)wtz_coop_bcast32((uint32_t)(acc[i] + (int32_t)wins[i].ovl));
		const int32_t qe = (int32_t)wtz_coop_bcast32((uint32_t)wins[i].end[0]), te = (int32_t)wtz_coop_bcast32((uint32_t)wins[i].end[1]);
		if(wi > top){ top = wi; top_at = (int32_t)i; }
		for(uint32_t j0 = i + 1; j0 < n; j0 += WTZ_NLANES){
			const uint32_t j = j0 + lane; const bool mine = j < n;
			int32_t g0 = 0, g1 = 0;
			if(mine){ g0 = wins[j].beg[0] - qe; g1 = wins[j].beg[1] - te; }
			const unsigned long long far = wtz_coop_ballot(mine && g0 > W && g1 > W);
			const uint32_t stop_at = far ? wtz_coop_first_lane(far) : WTZ_NLANES;
			if(mine && lane < stop_at && g0 >= 0 && g1 >= 0){
				const int32_t shift = g0 > g1 ? g0 - g1 : g1 - g0;
				if(shift <= W){
					const int32_t offer = wi - (int32_t)((float)shift * shift_cost);
					if(acc[j] < offer){ acc[j] = offer; pred[j] = (int32_t)i; }
				}
			}
			if(far) break;
		}
		WTZ_WAVE_SYNC();                                         /* the updates of this step are visible to the uniform reads of the next */
	}
	int32_t span = 0;
	for(int32_t k = top_at; k >= 0; k = (int32_t)wtz_coop_bcast32((uint32_t)pred[k])){
		if(lane == 0) wins[k].closed = 0;
		span += (int32_t)wtz_coop_bcast32((uint32_t)(wins[k].end[0] - wins[k].beg[0]));
	}
	WTZ_WAVE_SYNC();
	return span;
}

/* the chain members moved to the front of the array in their order (64 at a time: every lane reads its window before any lane writes; a group writes at or in
 * front of its own first slot, never into a group that has not been read); returns their number (uniform) */
WTZ_HD uint32_t wtz_keep_chain_members(wtz_win_t *wins, uint32_t n){
	const uint32_t lane = WTZ_LANE;
	uint32_t k = 0;
	for(uint32_t j0 = 0; j0 < n; j0 += WTZ_NLANES){
		const uint32_t j = j0 + lane;
		wtz_win_t w; bool keep = false;
		if(j < n){ w = wins[j]; keep = !w.closed; }
		uint32_t tot; const uint32_t at = wtz_coop_rank(keep, &tot);
		WTZ_WAVE_SYNC();
		if(keep) wins[k + at] = w;
		k += tot;
		WTZ_WAVE_SYNC();
	}
	return k;
}

/* result of one (query, candidate) pair; the window/anchor arrays live in the pool until the next stage reset */
typedef struct {
	uint32_t n_hits;                 /* |cache| (hzm_aln.h:173) */
	uint32_t gate;                   /* 1 if n_hits*zsize >= ztot (wtzmo.c:857) */
	uint32_t ovl[2];                 /* SEED[dir].ovl after chaining */
	uint32_t nwin[2];                /* chain windows kept (0 unless ovl >= ztot) */
	wtz_win_t  *win[2];
	wtz_zhit_t *anchors[2];
	uint32_t nanchors[2];
	int32_t  bad;                    /* pool exhausted while working on this pair */
	/* dot-matrix engine result (A7d) */
	int32_t dm_score, dm_qb, dm_qe, dm_tb, dm_te, dm_dir;
	uint32_t tick[4];               /* shader-clock ticks / 1024 spent in: matching, exact sort, windows+chain, total (profiling aid) */
} wtz_pairres_t;

#endif
