/*
 * wtz_tasks.h — the per-pair and per-window tasks launched by the stage functions of the C-ABI.
 *
 *   wtz_task_pair      one (query, candidate): A6 z-mer matches -> gate (wtzmo.c:857) ->
 *                      zmo: exact (off1,off2) order + windows + chain per strand (wtzmo.c:887-914)
 *                      dmo: dot-matrix alignment (wtzmo.c:858-886)
 *   wtz_task_winalign  one chain window: seed-anchored alignment A9 (wtzmo.c:1019-1028)
 *   wtz_task_stitch    one (pair, strand): stitch windows A10 (wtzmo.c:1030, hzm_aln.h:1345-1486)
 */
#ifndef WTZ_TASKS_H
#define WTZ_TASKS_H

#include "wtz_sw.h"
#include "wtz_sw_wave.h"
#include "wtz_sw_grp.h"
#include "wtz_sw_lane.h"
#include "wtz_dotmatrix.h"

typedef struct {
	wtz_reads_t R; wtz_zindex_t Z; const wtz_params_t *P; wtz_pool_t *pool;
	wtz_zindex_t ZQ;             /* the index the QUERY's z-mer table is read from: == Z unless the caller keeps the queries of the batch in a second index
	                              * (wtz_zindex_build_queries: several GPUs, each holding the candidate side of its own share of the reads only) */
	uint32_t dm_first_big;       /* dmo: a strand whose image does not fit the slice of the first K_pair launch keeps it in the pool instead of leaving the pair to a later launch */
} wtz_env_t;

#define WTZ_WAVE_LDS_BYTES 8192
#ifndef WTZ_GAP_LDS_BYTES
#define WTZ_GAP_LDS_BYTES 12288      /* LDS slice of a gap-filling (K-sw2) wave */
#endif
#ifndef WTZ_PAIR_LDS_BYTES
#define WTZ_PAIR_LDS_BYTES 8192      /* K_pair, zmo: LDS slice of the window scans (measured with WTZ_OCC_PAIR: 16 KB / 1 -> 117 ms, 8 KB / 3 -> 99, 8 KB / 5 -> 90) */
#endif
#ifndef WTZ_PAIR_DM_LDS_BYTES
#define WTZ_PAIR_DM_LDS_BYTES 24576  /* K_pair, dmo: LDS slice of the strand images (six waves per CU).  A strand that does not fit keeps its image in the pool
                                      * (dm_first_big) and pays an L2 round trip for everything the LDS form reads from the slice: at 20 KB that was 42 % of the
                                      * strands of configs[2] and 60 % of the denoise wave time (round-4 phase profile).  Round-4 sweep of the configs[2] dmo step,
                                      * 20 / 24 / 28 / 32 KB: K_pair_dm 5 735 / 5 220 / 5 423 / 5 462 ms.  (Round 3, when the per-band work dominated both forms alike,
                                      * measured 18 / 20 / 22 / 24 / 32 KB = 12.19 / 12.17 / 12.82 / 12.69 / 15.32 s and kept 20 KB.) */
#endif
/* the LDS slice of the wave running the current task (wave-task kernels carry WTZ_WAVE_LDS_BYTES of dynamic LDS) */
#if defined(__HIP_DEVICE_COMPILE__)
WTZ_D int32_t *wtz_wave_scratch(){ extern __shared__ int32_t wtz_dyn_lds[]; return wtz_dyn_lds; }
#else
WTZ_COOP_HOST int32_t *wtz_wave_scratch(){ return NULL; }
#endif

/* -DWTZ_DEBUG_CRUMBS: every pair task leaves its last reached point in a host-visible array, so that a hung or faulting K_pair launch
 * can be located from the host (WTZ_DEBUG_CRUMBS=1 makes wtz_pairs_seed poll instead of waiting and report the stragglers) */
#if defined(WTZ_DEBUG_CRUMBS) && defined(__HIPCC__)
__device__ unsigned int *wtz_crumbs = NULL;
#endif
#if defined(WTZ_DEBUG_CRUMBS) && defined(__HIP_DEVICE_COMPILE__)
#define WTZ_CRUMB(t, code) do { if(wtz_crumbs && WTZ_LANE == 0) __hip_atomic_store(&wtz_crumbs[t], (unsigned int)(code), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while(0)
#else
#define WTZ_CRUMB(t, code) do { } while(0)
#endif
/* all lanes of the wavefront enter; the z-mer matching is cooperative, the order-sensitive remainder runs on lane 0 */
/* ENGINE: 0 = windows + chain (zmo), 1 = dot matrix (dmo), -1 = decided at run time (host emulation).  One kernel per engine (round 3): a zmo launch does not
 * carry the registers and code of the dot-matrix path and vice versa. */
/* ZBIG (zmo): the window scans may use a workspace in the pool when a range does not fit the LDS slice.  The first launch runs without that body (its registers
 * and code would be carried by every pair) and marks the few pairs that need it (dm_dir = WTZ_PAIR_NEEDS_ZBIG); a second launch finishes them. */
#define WTZ_PAIR_NEEDS_ZBIG (-3)
template<int ENGINE = -1, bool ZBIG = true>
WTZ_HD void wtz_task_pair(uint32_t t, const wtz_env_t &V, const uint32_t *qid, const uint32_t *cid, wtz_pairres_t *res){
	const wtz_params_t *P = V.P;
	const bool dm = ENGINE < 0 ? (P->dot_matrix != 0) : (ENGINE == 1);
	const uint32_t q = qid[t], c = cid[t];
	wtz_pairres_t r; memset(&r, 0, sizeof r);
	wtz_zhit_t *hits = NULL; uint32_t n = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#define WTZ_TICK() ((uint64_t)clock64())
#else
#define WTZ_TICK() ((uint64_t)0)
#endif
	const uint64_t tk0 = WTZ_TICK();
	WTZ_CRUMB(t, 1);
	const bool aux = P->aux_strand != 0;                     /* align_hzmaux's form of the pair stages (wtgbo): strand 0 only, no n_hits gate */
#if defined(__HIP_DEVICE_COMPILE__)
	const bool ok0 = wtz_zmatch_coop(V.ZQ, V.Z, q, c, V.R.rdlen[c], P->max_kmer_var, V.pool, &hits, &n, aux, (uint32_t*)wtz_wave_scratch(), dm ? (uint32_t)WTZ_PAIR_DM_LDS_BYTES : (uint32_t)WTZ_PAIR_LDS_BYTES);      /* the LDS slice is free until the first ordering */
#else
	const bool ok0 = wtz_zmatch_coop(V.ZQ, V.Z, q, c, V.R.rdlen[c], P->max_kmer_var, V.pool, &hits, &n, aux);
#endif
	/* the same in every lane by construction; said explicitly, the early returns below are scalar branches and the window merge after them is not a masked region */
	const bool ok = wtz_coop_bcast32(ok0 ? 1u : 0u) != 0;
	n = wtz_coop_bcast32(n);
	hits = (wtz_zhit_t*)(uintptr_t)wtz_coop_bcast64((uint64_t)(uintptr_t)hits);
	const uint64_t tk1 = WTZ_TICK();
	WTZ_CRUMB(t, 2 | (n << 8));
	wtz_zhit_t *sorted = NULL;
#if defined(__HIP_DEVICE_COMPILE__)
	__threadfence_block();
	if(ok && (aux ? n > 0 : n * P->zsize >= P->ztot)){      /* uniform: the first ordering of either engine, wave-parallel when tie-free */
		int pbad = 0;
		if(dm) sorted = wtz_sort_hits_wave<1>(hits, n, V.pool, (uint64_t*)wtz_wave_scratch(), WTZ_PAIR_DM_LDS_BYTES / 8, &pbad);
		else              sorted = wtz_sort_hits_wave<0>(hits, n, V.pool, (uint64_t*)wtz_wave_scratch(), WTZ_PAIR_LDS_BYTES / 8, &pbad);
		if(wtz_coop_bcast32((uint32_t)pbad)) r.bad = 1;
		sorted = (wtz_zhit_t*)(uintptr_t)wtz_coop_bcast64((uint64_t)(uintptr_t)sorted);
	}
#endif
	const uint32_t lane = WTZ_LANE;
	WTZ_CRUMB(t, 3 | (n << 8));
	if(!ok || r.bad){ r.bad = 1; if(lane == 0) res[t] = r; WTZ_CRUMB(t, 0xFF); return; }
	r.n_hits = n;
	if(aux ? n == 0 : n * P->zsize < P->ztot){ r.gate = 0; if(lane == 0) res[t] = r; WTZ_CRUMB(t, 0xFF); return; }
	r.gate = 1;
	if(dm){
		wtz_vec<wtz_zhit_t> cache; cache.a = sorted ? sorted : hits; cache.n = n; cache.cap = n + 2; cache.pool = V.pool; cache.bad = 0;
		const uint64_t tk2 = WTZ_TICK();
		uint64_t tkd = 0;
#if defined(__HIP_DEVICE_COMPILE__)
		uint8_t *dlds = (uint8_t*)wtz_wave_scratch();
#else
		static thread_local uint64_t emul_dm_lds[WTZ_PAIR_DM_LDS_BYTES / 8];
		uint8_t *dlds = (uint8_t*)emul_dm_lds;
#endif
		wtz_dm_result_t d = wtz_dot_matrix_align(cache, V.pool, (int32_t)V.R.rdlen[q], (int32_t)V.R.rdlen[c], P, &r.bad, sorted != NULL, &tkd, dlds, WTZ_PAIR_DM_LDS_BYTES, true, V.dm_first_big != 0);
		if(lane != 0) return;
		if(d.dir == -2){ r.anchors[0] = cache.a; r.nanchors[0] = cache.n; }       /* deferred: the ordered matches stay in the pool for wtz_task_pair_dm_big */
		r.dm_score = d.score; r.dm_qb = d.qb; r.dm_qe = d.qe; r.dm_tb = d.tb; r.dm_te = d.te; r.dm_dir = d.dir;
		{ const uint64_t tk3 = WTZ_TICK(); r.tick[0] = (uint32_t)((tk1 - tk0) >> 10); r.tick[1] = (uint32_t)((tk2 - tk1) >> 10); r.tick[2] = (uint32_t)((tkd - tk2) >> 10); r.tick[3] = (uint32_t)((tk3 - tk0) >> 10); }   /* dmo: [2] = denoise */
		res[t] = r; return;
	}
	/* zmo: every lane follows the window merge (uniform control flow); the vectors belong to lane 0 */
	if(sorted) hits = sorted;                                                             /* tie-free: the unique ascending order */
	else { WTZ_CRUMB(t, 4 | (n << 8)); if(lane == 0) wtz_sort_exact(hits, (size_t)n, wtz_gt_off12()); WTZ_WAVE_SYNC(); }   /* process_hzmps, hzm_aln.h:1184-1186, swap-exact */
	WTZ_CRUMB(t, 5 | (n << 8));
	const uint64_t tk2 = WTZ_TICK();
	wtz_winscratch_t sc;
	{
		uint64_t pa = 0;
		if(lane == 0) pa = (uint64_t)(uintptr_t)wtz_pool_alloc(V.pool, (size_t)(n + 2) * (4 * 7 + 8 + sizeof(wtz_zhit_t)) + 16);
		pa = wtz_coop_bcast64(pa);
		sc.ts = (uint32_t*)(uintptr_t)pa;
	}
	if(sc.ts == NULL){ r.bad = 1; if(lane == 0) res[t] = r; WTZ_CRUMB(t, 0xFF); return; }
	sc.as = (int32_t*)(sc.ts + (n + 2)); sc.wb = sc.ts + 2 * (n + 2); sc.we = sc.ts + 3 * (n + 2); sc.wo = sc.ts + 4 * (n + 2); sc.wf = sc.ts + 5 * (n + 2); sc.wd = sc.ts + 6 * (n + 2);
	sc.tk = (uint64_t*)(sc.ts + 7 * (n + 2) + ((7 * (n + 2)) & 1)); sc.ztmp = (wtz_zhit_t*)(sc.tk + (n + 2));
#if defined(__HIP_DEVICE_COMPILE__)
	sc.lds = (uint64_t*)wtz_wave_scratch();
#else
	static thread_local uint64_t emul_lds[WTZ_PAIR_LDS_BYTES / 8];     /* host emulation: the LDS slice of the (single-lane) wave */
	sc.lds = emul_lds;
#endif
	sc.lds_u64 = WTZ_PAIR_LDS_BYTES / 8;
	sc.big = NULL; sc.big_u64 = 0; sc.need_big = 0;
#ifndef WTZ_MERGE_SCALAR
	wtz_merge_prepare(hits, n, sc, P->kwin);
#endif
	for(uint32_t dir = 0; dir < 2; dir++){
		wtz_vec<wtz_win_t> wins; wtz_vec<wtz_zhit_t> anchors;
		wins.a = NULL; wins.n = wins.cap = 0; wins.pool = V.pool; wins.bad = 0;
		anchors.a = NULL; anchors.n = anchors.cap = 0; anchors.pool = V.pool; anchors.bad = 0;
		if(lane == 0){ wins.init(V.pool, 16); anchors.init(V.pool, n + 16); }
		WTZ_CRUMB(t, (6 + dir) | (n << 8));
		const unsigned long long ptm = WTZ_PROF_T(); (void)ptm;
#ifdef WTZ_MERGE_SCALAR
		const uint32_t nw = wtz_merge_windows_coop<ZBIG>(hits, n, dir, wins, anchors, sc, P->zsize, P->kwin, P->kstep, P->zovl);
#else
		const uint32_t nw = wtz_merge_windows_wave<ZBIG>(hits, n, dir, wins, anchors, sc, P->zsize, P->kwin, P->kstep, P->zovl);
#endif
		WTZ_PROF_ADD(63, ptm);      /* profiler: 63 = merge loop incl. its scans, 62 = chain + compaction */
		if(!ZBIG && sc.need_big){          /* a range that does not fit the LDS slice: the whole pair again in the launch with the pool-workspace body */
			if(lane == 0){ wtz_pairres_t r2; memset(&r2, 0, sizeof r2); r2.n_hits = r.n_hits; r2.gate = 1; r2.dm_dir = WTZ_PAIR_NEEDS_ZBIG; res[t] = r2; }
			WTZ_CRUMB(t, 0xFF);
			return;
		}
		WTZ_CRUMB(t, (8 + dir) | (n << 8));
		/* the vectors belong to lane 0: what the chain needs of them as uniform values */
		if(wtz_coop_bcast32((uint32_t)(wins.bad | anchors.bad))){ r.bad = 1; continue; }
		if(nw == 0) continue;
		const unsigned long long ptc = WTZ_PROF_T(); (void)ptc;
		WTZ_WAVE_SYNC();                                       /* lane 0 wrote the windows; every lane reads them */
		const uint32_t wn = wtz_coop_bcast32(wins.n);
		wtz_win_t *wa = (wtz_win_t*)(uintptr_t)wtz_coop_bcast64((uint64_t)(uintptr_t)wins.a);
		int32_t *mem = NULL;
		if(wn > 64u || WTZ_NLANES == 1u){
			uint64_t pa = 0;
			if(lane == 0) pa = (uint64_t)(uintptr_t)wtz_pool_alloc(V.pool, (size_t)wn * 8 + 8);
			mem = (int32_t*)(uintptr_t)wtz_coop_bcast64(pa);
			if(mem == NULL){ r.bad = 1; continue; }
		}
		r.ovl[dir] = WTZ_OVL29(wtz_chain_windows_wave(wa, wn, P->W, mem));
		if(r.ovl[dir] < P->ztot) continue;
		/* keep the chain members in place (compacted to the front); their anchors[] keep indexing `anchors` */
		const uint32_t k = wtz_keep_chain_members(wa, wn);
		r.nwin[dir] = k; r.win[dir] = wa; r.anchors[dir] = anchors.a; r.nanchors[dir] = anchors.n;
		WTZ_PROF_ADD(62, ptc);
	}
	if(lane != 0) return;
	{ const uint64_t tk3 = WTZ_TICK(); r.tick[0] = (uint32_t)((tk1 - tk0) >> 10); r.tick[1] = (uint32_t)((tk2 - tk1) >> 10); r.tick[2] = (uint32_t)((tk3 - tk2) >> 10); r.tick[3] = (uint32_t)((tk3 - tk0) >> 10); }
	res[t] = r;
	WTZ_CRUMB(t, 0xFF);
}

/* dmo pairs whose strand images did not fit the LDS slice of K_pair: same alignment over the already ordered matches, launched
 * with another slice (`lds_bytes`); `big` = the strand image may live in the pool (wtz_denoise_dir_coop); `last` = no further launch
 * follows, so nothing is deferred again */
WTZ_HD void wtz_task_pair_dm_big(uint32_t t, const wtz_env_t &V, const uint32_t *list, const uint32_t *qid, const uint32_t *cid, wtz_pairres_t *res, uint32_t lds_bytes, bool last, bool big){
	const wtz_params_t *P = V.P;
	const uint32_t pi = list[t], q = qid[pi], c = cid[pi];
	wtz_pairres_t r = res[pi];
	wtz_vec<wtz_zhit_t> cache; cache.a = r.anchors[0]; cache.n = r.nanchors[0]; cache.cap = cache.n + 2; cache.pool = V.pool; cache.bad = 0;
	uint64_t tkd = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	uint8_t *dlds = (uint8_t*)wtz_wave_scratch();
	wtz_dm_result_t d = wtz_dot_matrix_align(cache, V.pool, (int32_t)V.R.rdlen[q], (int32_t)V.R.rdlen[c], P, &r.bad, true, &tkd, dlds, lds_bytes, !last, big);
#else
	(void)lds_bytes; (void)last; (void)big;
	wtz_dm_result_t d = wtz_dot_matrix_align(cache, V.pool, (int32_t)V.R.rdlen[q], (int32_t)V.R.rdlen[c], P, &r.bad, true, &tkd, (uint8_t*)NULL, 0, false, false);
#endif
	if(WTZ_LANE != 0) return;
	r.dm_score = d.score; r.dm_qb = d.qb; r.dm_qe = d.qe; r.dm_tb = d.tb; r.dm_te = d.te; r.dm_dir = d.dir;
	res[pi] = r;
}

/* ---------------- alignment ---------------- */
typedef struct {                 /* one chain window to align */
	uint32_t item;               /* index of the (pair,strand) item it belongs to */
	uint32_t widx;               /* window index inside the pair's chain */
} wtz_wintask_t;

typedef struct { wtz_aln_t x; uint32_t *cigar; uint32_t cigar_len; uint32_t pass; unsigned long long cells; } wtz_reg_t;   /* aln_reg_t + pass flag (wtzmo.c:1026) */

typedef struct {                 /* one (pair,strand) to align */
	uint32_t q, c, dir;
	const wtz_win_t *win; const wtz_zhit_t *anchors; uint32_t nwin;
	wtz_reg_t *regs;             /* nwin entries, filled by wtz_task_winalign */
} wtz_alnitem_t;

typedef struct {
	wtz_aln_t x; uint32_t n_regs; uint32_t cigar_len; uint32_t *cigar; int32_t bad;
	unsigned long long cells_shift, cells_fixed, cells_global;
	uint32_t text_len;
} wtz_alnres_dev_t;

WTZ_HD uint32_t wtz_cigar_text_len(const uint32_t *c, uint32_t n){       /* kswx.h:1093-1120: zero-length ops are skipped */
	uint32_t k = 0;
	for(uint32_t i = 0; i < n; i++){ uint32_t len = c[i] >> 4; if(len == 0) continue; uint32_t d = 1; while(len >= 10){ len /= 10; d++; } k += d + 1; }
	return k;
}
WTZ_HD void wtz_cigar_text_write(const uint32_t *c, uint32_t n, char *s){
	uint32_t k = 0;
	for(uint32_t i = 0; i < n; i++){
		uint32_t op = c[i] & 0xF, len = c[i] >> 4;
		if(len == 0) continue;
		char d[12]; int nd = 0;
		while(len){ d[nd++] = (char)('0' + len % 10); len /= 10; }
		while(nd) s[k++] = d[--nd];
		s[k++] = op == 0 ? 'M' : (op == 1 ? 'I' : 'D');
	}
}

/* ---- the same CIGAR plumbing with the whole wavefront (all lanes call with identical arguments; vectors are kept
 *      identical on every lane, storage comes from lane 0).  Lists are a few thousand operations per overlap: looping over
 *      them on one lane costs a dependent HBM access per operation. ---- */
WTZ_HD bool wtz_cigar_reserve_coop(wtz_cigar_t &c, uint32_t want){
	if(want <= c.cap) return true;
	uint32_t cap = c.cap ? c.cap : 16; while(cap < want) cap <<= 1;
	uint64_t a = 0;
	if(WTZ_LANE == 0) a = (uint64_t)(uintptr_t)wtz_pool_alloc(c.pool, (size_t)cap * 4);
	a = wtz_coop_bcast64(a);
	uint32_t *b = (uint32_t*)(uintptr_t)a;
	if(b == NULL){ c.bad = 1; return false; }
	for(uint32_t i = WTZ_LANE; i < c.n; i += WTZ_NLANES) b[i] = c.a[i];
	WTZ_WAVE_SYNC();
	c.a = b; c.cap = cap; return true;
}
WTZ_HD void wtz_cigar_concat_coop(wtz_cigar_t &c, const uint32_t *src, uint32_t n){      /* kswx.h:46-52 */
	if(n == 0) return;
	uint32_t k0 = 0;
	if(c.n && (c.a[c.n - 1] & 0xFu) == (src[0] & 0xFu)){ if(WTZ_LANE == 0) c.a[c.n - 1] += src[0] & 0xFFFFFFF0u; k0 = 1; }
	if(!wtz_cigar_reserve_coop(c, c.n + n)) return;
	for(uint32_t k = k0 + WTZ_LANE; k < n; k += WTZ_NLANES) c.a[c.n + k - k0] = src[k];
	c.n += n - k0;
	WTZ_WAVE_SYNC();
}
WTZ_HD void wtz_cigar_reverse_coop(uint32_t *a, uint32_t n){
	for(uint32_t i = WTZ_LANE; i < n / 2; i += WTZ_NLANES){ const uint32_t t = a[i]; a[i] = a[n - 1 - i]; a[n - 1 - i] = t; }
	WTZ_WAVE_SYNC();
}
/* Several operation lists appended to `c` at once (what a sequence of wtz_cigar_concat_coop calls gives, kswx.h:46-52): a list whose first operation equals the last
 * operation before it loses that element to its predecessor, every other element is copied.  One pass over the descriptors (where each list goes, whether it merges)
 * instead of a dependent load + reserve + copy + fence per list: a stitched overlap is 2 x windows lists of a few dozen operations each.
 * piece(p, &src, &n, &rev) describes list p of np (src may be read back to front: the left extension's list).  `lds` holds 6 words per list. */
template<typename PF>
WTZ_HD bool wtz_cigar_join_coop(wtz_cigar_t &c, uint32_t np, PF piece, uint32_t *lds, uint32_t lds_words){
	if((uint64_t)np * 6u > lds_words) return false;
	const uint32_t lane = WTZ_LANE;
	uint32_t *LB = lds, *LN = lds + np, *LF = lds + 2 * np, *LW = lds + 3 * np; uint64_t *LS = (uint64_t*)(lds + 4 * np);      /* 4 np is even: 8-byte aligned */
	uint32_t carry_valid = c.n ? 1u : 0u, carry_last = c.n ? c.a[c.n - 1] : 0u, total = 0;
	for(uint32_t p0 = 0; p0 < np; p0 += WTZ_NLANES){
		const uint32_t p = p0 + lane;
		const uint32_t *src = NULL; uint32_t n = 0, rev = 0, fw = 0, lw = 0;
		if(p < np){ piece(p, &src, &n, &rev); if(n){ fw = rev ? src[n - 1] : src[0]; lw = rev ? src[0] : src[n - 1]; } }
		const unsigned long long mask = wtz_coop_ballot(n != 0);
		const unsigned long long below = mask & ((1ull << lane) - 1ull);
		const uint32_t from = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
		const uint32_t plw = wtz_coop_shfl32(lw, from);
		const uint32_t prev = below ? plw : carry_last; const bool prev_ok = below ? true : (carry_valid != 0);
		const uint32_t merged = (n && prev_ok && (fw & 0xFu) == (prev & 0xFu)) ? 1u : 0u;
		uint32_t tot; const uint32_t ex = wtz_coop_excl_scan(n - merged, &tot);
		if(p < np){ LB[p] = total + ex; LN[p] = n; LF[p] = merged | (rev << 1); LW[p] = fw; LS[p] = (uint64_t)(uintptr_t)src; }
		if(mask){ const uint32_t top = 63u - (uint32_t)__builtin_clzll(mask); carry_last = wtz_coop_lane32(lw, top); carry_valid = 1; }
		total += tot;
	}
	WTZ_WAVE_SYNC();
	const uint32_t n0 = c.n;
	if(!wtz_cigar_reserve_coop(c, n0 + total)) return true;          /* c.bad is set */
	uint32_t *dst = c.a + n0;
	for(uint32_t e = lane; e < total; e += WTZ_NLANES){
		uint32_t lo = 0, hi = np - 1;                                     /* the last list whose first output position is <= e */
		while(lo < hi){ const uint32_t mid = (lo + hi + 1) >> 1; if(LB[mid] <= e) lo = mid; else hi = mid - 1; }
		const uint32_t fl = LF[lo], n = LN[lo], k = e - LB[lo] + (fl & 1u);
		const uint32_t *src = (const uint32_t*)(uintptr_t)LS[lo];
		dst[e] = (fl & 2u) ? src[n - 1 - k] : src[k];
	}
	WTZ_WAVE_SYNC();
	for(uint32_t p = lane; p < np; p += WTZ_NLANES) if(LF[p] & 1u){
		const uint32_t add = LW[p] & 0xFFFFFFF0u;                         /* lists of one element can pile up on the same predecessor */
		uint32_t *w = dst + LB[p]; w--;                                   /* the element in front of the list's first output position (c's old last element for position 0) */
#if defined(__HIP_DEVICE_COMPILE__)
		atomicAdd(w, add);
#else
		*w += add;
#endif
	}
	c.n = n0 + total;
	WTZ_WAVE_SYNC();
	return true;
}
WTZ_HD uint32_t wtz_cigar_text_len_coop(const uint32_t *c, uint32_t n){
	uint32_t tot = 0;
	for(uint32_t i0 = 0; i0 < n; i0 += WTZ_NLANES){
		const uint32_t i = i0 + WTZ_LANE;
		uint32_t k = 0;
		if(i < n){ uint32_t len = c[i] >> 4; if(len){ uint32_t d = 1; while(len >= 10){ len /= 10; d++; } k = d + 1; } }
		uint32_t t; (void)wtz_coop_excl_scan(k, &t); tot += t;
	}
	return tot;
}
WTZ_HD void wtz_cigar_text_write_coop(const uint32_t *c, uint32_t n, char *s){
	uint32_t base = 0;
	for(uint32_t i0 = 0; i0 < n; i0 += WTZ_NLANES){
		const uint32_t i = i0 + WTZ_LANE;
		uint32_t k = 0, op = 0, len = 0;
		if(i < n){ op = c[i] & 0xFu; len = c[i] >> 4; if(len){ uint32_t l2 = len, d = 1; while(l2 >= 10){ l2 /= 10; d++; } k = d + 1; } }
		uint32_t t; const uint32_t ex = wtz_coop_excl_scan(k, &t);
		if(k){
			char *o = s + base + ex; uint32_t p = k - 1;
			o[p] = op == 0 ? 'M' : (op == 1 ? 'I' : 'D');
			while(p){ o[--p] = (char)('0' + len % 10); len /= 10; }
		}
		base += t;
	}
}

WTZ_HD wtz_readview wtz_view(const wtz_reads_t &R, uint32_t id, uint32_t rev){ wtz_readview v; v.bits = R.bits; v.off = R.rdoff[id]; v.len = R.rdlen[id]; v.rev = rev; return v; }

/* launched wave-cooperatively: on the GPU the K-sw1 gaps between anchors are computed by the whole wavefront
 * (wtz_align_window_wave), the host emulation runs the scalar body */
template<bool FULL = true>
WTZ_HD void wtz_task_winalign(uint32_t t, const wtz_env_t &V, const wtz_wintask_t *tasks, const wtz_alnitem_t *items, uint32_t *defer = NULL, const uint32_t *list = NULL){
	if(list) t = list[1 + t];
	const wtz_params_t *P = V.P;
	const wtz_alnitem_t &it = items[tasks[t].item];
	const wtz_win_t &w = it.win[tasks[t].widx];
	wtz_reg_t reg; memset(&reg, 0, sizeof reg);
	wtz_cigar_t cigar, tmp;
	/* the window's CIGAR is sized from its anchor count (about a dozen runs per anchor): growing it by doubling copied every element
	 * on lane 0 two or three times per window; tmp only serves the scalar body */
	cigar.init(V.pool, WTZ_LANE == 0 ? (w.anchors[1] - w.anchors[0]) * 14u + 16u : 0); tmp.init(V.pool, 0);
	unsigned long long cells = 0;
	int32_t bad = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	bool ok = true, dfr = false;
	reg.x = wtz_align_window_wave<FULL>(wtz_view(V.R, it.q, 0), wtz_view(V.R, it.c, it.dir), w, it.anchors, cigar, tmp, P, V.pool, wtz_wave_scratch(), &cells, &ok, &dfr);
	if(!FULL && dfr){ if(WTZ_LANE == 0){ const uint32_t idx = atomicAdd(&defer[0], 1u); defer[1 + idx] = t; } return; }
	if(!ok) bad = 1;
	if(WTZ_LANE != 0) return;
#else
	wtz_swmem_t mem; wtz_swmem_init_lds(mem, V.pool, wtz_wave_scratch(), WTZ_WAVE_LDS_BYTES / 4);
	reg.x = wtz_align_window(wtz_view(V.R, it.q, 0), wtz_view(V.R, it.c, it.dir), w, it.anchors, cigar, mem, tmp, P);
	if(mem.bad) bad = 1;
#endif
	reg.cigar = cigar.a; reg.cigar_len = cigar.n; reg.cells = cells;
	reg.pass = !(reg.x.aln * 2 < (int32_t)P->zovl || (float)reg.x.mat < (float)reg.x.aln * P->min_id);
	if(cigar.bad || tmp.bad || bad) reg.pass = 2;        /* pool exhausted */
	it.regs[tasks[t].widx] = reg;
}

/* ---------------- A9 with one lane per K-sw1 problem (wtz_sw_lane.h): planner, DP, fold ---------------- */
#if defined(__HIP_DEVICE_COMPILE__)
#define WTZ_ATOMIC_INC32(p) atomicAdd((p), 1u)
#define WTZ_ATOMIC_ADD64(p, v) atomicAdd((p), (unsigned long long)(v))
#else
#define WTZ_ATOMIC_INC32(p) ((*(p))++)
#define WTZ_ATOMIC_ADD64(p, v) (*(p) += (unsigned long long)(v))
#endif
/* window t: number of anchors = number of problem slots */
WTZ_HD void wtz_task_lcount(uint32_t t, const wtz_wintask_t *tasks, const wtz_alnitem_t *items, uint32_t *na){
	const wtz_alnitem_t &it = items[tasks[t].item];
	const wtz_win_t &w = it.win[tasks[t].widx];
	na[t] = w.anchors[1] - w.anchors[0];
}
/* The planner walks the anchors of window t like hzm_aln.h:1255-1300 does, without any DP: the gap before a usable anchor ends ON the anchor
 * (hzm_aln.h:1273-1284 patch whatever the extension left), and the z-mer run alignment moves the cursor by sums of run lengths.  Slot k of
 * the window = its k-th anchor: the K-sw1 problem in front of it (qlen < 0: none), the room its run list needs, and the sort key
 * 0xFFFF - (n_col << 9 | rows) (descending shape order; 0xFFFF = no DP).  wflag = 1: the window has a problem outside the lane envelope
 * (or whose band depends on init_score: only with -w above 48 + 2 min(qlen, tlen)) and is left to the chained kernel. */
WTZ_HD void wtz_task_lplan(uint32_t t, const wtz_env_t &V, const wtz_wintask_t *tasks, const wtz_alnitem_t *items, const uint32_t *woff,
		wtz_lprob_t *prob, uint32_t *runcap, uint64_t *key, uint32_t *val, uint8_t *wflag, uint32_t *ccnt, uint32_t *uidx, uint32_t *nu){
	const wtz_params_t *P = V.P;
	const wtz_alnitem_t &it = items[tasks[t].item];
	const wtz_win_t &w = it.win[tasks[t].widx];
	const wtz_readview pb1 = wtz_view(V.R, it.q, 0), pb2 = wtz_view(V.R, it.c, it.dir);
	const uint32_t base = woff[t], na = w.anchors[1] - w.anchors[0];
	const int32_t M = P->M, I = P->O, D = P->O, E = P->E, T = P->T;
	int32_t te = 0, qe = 0; bool first = true, fb = false, ended = false;
	uint32_t cc[4] = {0u, 0u, 0u, 0u}, n_used = 0;
	for(uint32_t k0 = 0; k0 < na; k0 += 4){
		/* four anchors at a time, and the z-mers of all four speculatively: these loads do not depend on the walk, only their use does (a lane
		 * uses about one anchor in five, and the lanes of a wave use theirs at different iterations: a load inside the walk costs its latency 64 times) */
		wtz_zhit_t A[4]; wtz_seq_w2 Z1[4], Z2[4];
		#pragma unroll
		for(uint32_t j = 0; j < 4; j++) if(k0 + j < na) A[j] = it.anchors[w.anchors[0] + k0 + j];
		#pragma unroll
		for(uint32_t j = 0; j < 4; j++){
			Z1[j].w0 = Z1[j].w1 = Z2[j].w0 = Z2[j].w1 = 0;
			if(k0 + j < na && !ended && ZH_LEN1(A[j]) <= 64 && ZH_LEN2(A[j]) <= 64){
				Z1[j] = wtz_seq_load_w2(pb1.sub((int32_t)ZH_OFF1(A[j]), 1), ZH_LEN1(A[j])); Z2[j] = wtz_seq_load_w2(pb2.sub((int32_t)ZH_OFF2(A[j]), 1), ZH_LEN2(A[j]));
			}
		}
		#pragma unroll
		for(uint32_t j = 0; j < 4; j++){
			const uint32_t k = k0 + j;
			if(k >= na) break;
			wtz_lprob_t pr; pr.win = t; pr.qoff = pr.toff = 0; pr.qlen = -1; pr.tlen = 0; pr.run_off = 0;
			uint32_t cap = 0; uint64_t ky = 0xFFFFull;
			if(!ended){
				const wtz_zhit_t p = A[j];
				const int32_t off1 = (int32_t)ZH_OFF1(p), off2 = (int32_t)ZH_OFF2(p);
				if(first){ te = off1; qe = off2; first = false; }
				if(off1 >= te && off2 >= qe){
					pr.qoff = qe; pr.toff = te; pr.qlen = off2 - qe; pr.tlen = off1 - te;
					uidx[base + n_used++] = k;
					if(pr.qlen > 0 && pr.tlen > 0){
						int32_t W0 = P->w, W1 = P->w, ql, tl, n_col, ql1, tl1, nc1;
						wtz_ext_geometry(pr.qlen, pr.tlen, 0, W0, M, I, D, E, T, ql, tl, n_col);
						wtz_ext_geometry(pr.qlen, pr.tlen, 1 << 24, W1, M, I, D, E, T, ql1, tl1, nc1);
						if(W0 != W1 || n_col > WTZ_LN_MAXCOLS || ql > WTZ_LN_MAXROWS || ql + tl > WTZ_LN_MAXSPAN) fb = true;
						else { cap = (uint32_t)(ql + tl + 2); ky = 0xFFFFull - (uint64_t)(((uint32_t)n_col << 9) | (uint32_t)ql); cc[wtz_lane_class(n_col)]++; }
					}
					int32_t dte = 0, dqe = 0; bool past = false, ok;
					const uint32_t len1 = ZH_LEN1(p), len2 = ZH_LEN2(p);
					if(len1 <= 64 && len2 <= 64) ok = wtz_zmer_advance_w2(Z1[j], len1, Z2[j], len2, &dte, &dqe, &past); else { ok = false; past = true; }
					if(past) ok = wtz_zmer_advance_slow(pb1.sub(off1, 1), len1, pb2.sub(off2, 1), len2, &dte, &dqe);
					if(!ok) ended = true;      /* hzm_aln.h:1288-1291: the window ends here */
					else { te = off1 + dte; qe = off2 + dqe; }
				}
			}
			prob[base + k] = pr; runcap[base + k] = cap; key[base + k] = ky; val[base + k] = base + k;
		}
	}
	nu[t] = n_used;
	wflag[t] = fb ? 1 : 0;
	if(fb){ for(uint32_t k = 0; k < na; k++){ key[base + k] = 0xFFFFull; runcap[base + k] = 0; } }
	/* the four class counters: summed over the wavefront first (round 6: up to four same-address atomics per LANE - tens of millions per configs[2] step; K_gplan lost 35
	 * of its 38 ms with the same change).  A wavefront that is not full (the last of the launch) keeps the per-lane form: a shuffle must not read a lane that is not there. */
#if defined(__HIP_DEVICE_COMPILE__)
	{
		const bool full = __ballot(1) == ~0ull;
		#pragma unroll
		for(int q4 = 0; q4 < 4; q4++){
			uint32_t v = fb ? 0u : cc[q4];
			if(full){
				#pragma unroll
				for(int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
				if((threadIdx.x & 63u) == 0u && v) atomicAdd(&ccnt[q4], v);
			} else if(v) atomicAdd(&ccnt[q4], v);
		}
	}
#else
	if(!fb) for(int q4 = 0; q4 < 4; q4++) ccnt[q4] += cc[q4];
#endif
}

/* one wavefront = WTZ_NLANES problems of the shape-sorted order[lo, hi): relative-mode K-sw1, forward part; the wave's trace rows stay in the
 * transient pool (wtr[gw] = their base, wrm[gw] = rows per lane) for the traceback kernel */
template<int NC>
WTZ_HD void wtz_task_ldp(uint32_t gw, uint32_t wv, const wtz_env_t &V, const wtz_wintask_t *tasks, const wtz_alnitem_t *items, const uint32_t *order, uint32_t lo, uint32_t hi,
		const wtz_lprob_t *prob, wtz_lres_t *res, uint64_t *wtr, uint32_t *wrm){
	const wtz_params_t *P = V.P;
	const uint32_t idx = lo + wv * WTZ_NLANES + WTZ_LANE;
	const bool live = idx < hi;
	const uint32_t slot = live ? order[idx] : 0u;
	wtz_lprob_t pr; pr.win = 0; pr.qoff = pr.toff = 0; pr.qlen = pr.tlen = 0; pr.run_off = 0;
	if(live) pr = prob[slot];
	const int32_t M = P->M, X = P->X, I = P->O, D = P->O, E = P->E, T = P->T;
	int32_t W = P->w, ql = 0, tl = 0, n_col = 0;
	wtz_seq_packed q, tt; q.bits = tt.bits = V.R.bits; q.start = tt.start = 0; q.strand = tt.strand = 1; q.comp = tt.comp = 0;
	if(live){
		wtz_ext_geometry(pr.qlen, pr.tlen, 0, W, M, I, D, E, T, ql, tl, n_col);
		const wtz_alnitem_t &it = items[tasks[pr.win].item];
		q = wtz_view(V.R, it.c, it.dir).sub(pr.qoff, 1); tt = wtz_view(V.R, it.q, 0).sub(pr.toff, 1);
	}
	constexpr uint32_t RS = (uint32_t)wtz_lane_geo<NC>::RS;
	const uint32_t rows_max = (uint32_t)wtz_lane_wmax(ql);
	uint64_t pa = 0;
	if(WTZ_LANE == 0){
		pa = (uint64_t)(uintptr_t)wtz_pool_alloc(V.pool + 1, (size_t)WTZ_NLANES * rows_max * RS * 4 + 16);      /* the wave's trace rows: transient pool */
		wtr[gw] = pa; wrm[gw] = rows_max;
	}
	pa = wtz_coop_bcast64(pa);
	if(pa == 0) return;                       /* the slots stay "not done": the fold reports the pool */
	uint32_t *tr = (uint32_t*)(uintptr_t)pa;      /* the wave's block; the lanes' rows are interleaved inside it (wtz_ltr_at) */
	wtz_lres_t R; memset(&R, 0, sizeof R);
	wtz_lane_fixed<NC, false>(live, pr.qlen, q, pr.tlen, tt, 0, W, ql, tl, M, X, I, D, E, T, tr, R);
	if(live) res[slot] = R;
}
/* traceback of the same wave: lane l walks the rows lane l wrote */
WTZ_HD void wtz_task_ltb(uint32_t gw, uint32_t wv, uint32_t RS, const wtz_env_t &V, const wtz_wintask_t *tasks, const wtz_alnitem_t *items, const uint32_t *order, uint32_t lo, uint32_t hi,
		const wtz_lprob_t *prob, const uint32_t *runoff, uint32_t *runs, wtz_lres_t *res, const uint64_t *wtr, const uint32_t *wrm){
	const wtz_params_t *P = V.P;
	const uint32_t idx = lo + wv * WTZ_NLANES + WTZ_LANE;
	const uint64_t pa = wtr[gw];
	if(idx >= hi || pa == 0) return;
	const uint32_t slot = order[idx];
	const wtz_lprob_t pr = prob[slot];
	wtz_lres_t R = res[slot];
	int32_t W = P->w, ql, tl, n_col;
	wtz_ext_geometry(pr.qlen, pr.tlen, 0, W, P->M, P->O, P->O, P->E, P->T, ql, tl, n_col);
	const wtz_alnitem_t &it = items[tasks[pr.win].item];
	const wtz_seq_packed q = wtz_view(V.R, it.c, it.dir).sub(pr.qoff, 1), tt = wtz_view(V.R, it.q, 0).sub(pr.toff, 1);
	const uint32_t *tr = (const uint32_t*)(uintptr_t)pa;      /* lane-interleaved, as the DP kernel wrote it */
	wtz_lane_traceback<false>(true, R.qe - 1, R.te - 1, W, RS, q, pr.qlen, tt, pr.tlen, tr, runs + runoff[slot], R);
	res[slot] = R;
}

/* the four shape classes in ONE launch (widest band first = longest tasks first): wave wv -> class by the launch's wave ranges */
typedef struct { uint32_t wend[4], lo[4], hi[4]; } wtz_lclass_t;      /* index 0..3 = band classes 104 / 64 / 32 / 16 */
WTZ_HD void wtz_task_ldp_all(uint32_t wv, const wtz_lclass_t &L, const wtz_env_t &V, const wtz_wintask_t *tasks, const wtz_alnitem_t *items, const uint32_t *order,
		const wtz_lprob_t *prob, wtz_lres_t *res, uint64_t *wtr, uint32_t *wrm){
	uint32_t k = 0; while(k < 3 && wv >= L.wend[k]) k++;
	const uint32_t wl = wv - (k ? L.wend[k - 1] : 0u);
	if(k == 0)      wtz_task_ldp<104>(wv, wl, V, tasks, items, order, L.lo[0], L.hi[0], prob, res, wtr, wrm);
	else if(k == 1) wtz_task_ldp<64>(wv, wl, V, tasks, items, order, L.lo[1], L.hi[1], prob, res, wtr, wrm);
	else if(k == 2) wtz_task_ldp<32>(wv, wl, V, tasks, items, order, L.lo[2], L.hi[2], prob, res, wtr, wrm);
	else            wtz_task_ldp<16>(wv, wl, V, tasks, items, order, L.lo[3], L.hi[3], prob, res, wtr, wrm);
}
WTZ_HD void wtz_task_ltb_all(uint32_t wv, const wtz_lclass_t &L, const wtz_env_t &V, const wtz_wintask_t *tasks, const wtz_alnitem_t *items, const uint32_t *order,
		const wtz_lprob_t *prob, const uint32_t *runoff, uint32_t *runs, wtz_lres_t *res, const uint64_t *wtr, const uint32_t *wrm){
	uint32_t k = 0; while(k < 3 && wv >= L.wend[k]) k++;
	const uint32_t wl = wv - (k ? L.wend[k - 1] : 0u);
	wtz_task_ltb(wv, wl, 16u >> k, V, tasks, items, order, L.lo[k], L.hi[k], prob, runoff, runs, res, wtr, wrm);      /* trace dwords per row: 16 / 8 / 4 / 2 */
}

/* lane-private CIGAR vector with the open run in a register */
typedef struct { wtz_cigar_t v; uint32_t tail; } wtz_lcig_t;
WTZ_HD void wtz_lcig_push(wtz_lcig_t &w, uint32_t op, uint32_t len){
	if(len == 0) return;
	if(w.tail && (w.tail & 0xFu) == op) w.tail += len << 4;
	else { if(w.tail) w.v.push(w.tail); w.tail = (len << 4) | op; }
}
/* hzm_aln.h:278-314 pushing into the lane writer; aln == 0: the pair does not align (the caller rolls the writer back).
 * *past = true: the loop would read beyond one of the two z-mers (only the base-by-base form can follow the reference there) */
template<typename S1, typename S2>
WTZ_HD wtz_aln_t wtz_align_zmer_lane_t(const S1 &pb1, uint32_t len1, const S2 &pb2, uint32_t len2, int32_t M, int32_t I, int32_t D, int32_t E, wtz_lcig_t &cw, bool bounded, bool *past){
	wtz_aln_t x, zero; memset(&zero, 0, sizeof zero); x = zero;
	uint32_t s0 = 0, s1 = 0;
	while(s0 < len1 || s1 < len2){
		if(bounded && (s0 >= len1 || s1 >= len2)){ *past = true; return zero; }
		const uint32_t b = pb1.at((int32_t)s0);
		if(b != pb2.at((int32_t)s1)) return zero;
		uint32_t e0 = s0 + 1; while(e0 < len1 && pb1.at((int32_t)e0) == b) e0++;
		uint32_t e1 = s1 + 1; while(e1 < len2 && pb2.at((int32_t)e1) == b) e1++;
		const uint32_t l0 = e0 - s0, l1 = e1 - s1;
		if(l0 < l1){ x.aln += l1; x.mat += l0; x.ins += l1 - l0; x.score += (int32_t)l0 * M + I + (int32_t)(l1 - l0) * E; wtz_lcig_push(cw, 0, l0); wtz_lcig_push(cw, 1, l1 - l0); }
		else if(l0 == l1){ x.aln += l0; x.mat += l0; x.score += (int32_t)l0 * M; wtz_lcig_push(cw, 0, l0); }
		else { x.aln += l0; x.mat += l1; x.del += l0 - l1; x.score += (int32_t)l1 * M + D + (int32_t)(l0 - l1) * E; wtz_lcig_push(cw, 0, l1); wtz_lcig_push(cw, 2, l0 - l1); }
		s0 = e0; s1 = e1;
	}
	x.te = x.mat + x.del; x.qe = x.mat + x.ins;
	return x;
}
WTZ_HD wtz_aln_t wtz_align_zmer_lane(const wtz_seq_packed &pb1, uint32_t len1, const wtz_seq_packed &pb2, uint32_t len2, int32_t M, int32_t I, int32_t D, int32_t E, wtz_lcig_t &cw){
	bool past = false;
	if(len1 <= 64 && len2 <= 64){
		const wtz_seq_w2 r1 = wtz_seq_load_w2(pb1, len1), r2 = wtz_seq_load_w2(pb2, len2);
		if(len1 == len2 && len1 && r1.w0 == r2.w0 && r1.w1 == r2.w1){        /* the same string: the runs merge into one M of the whole length */
			wtz_aln_t y; memset(&y, 0, sizeof y);
			y.aln = y.mat = (int32_t)len1; y.te = y.qe = (int32_t)len1; y.score = (int32_t)len1 * M;
			wtz_lcig_push(cw, 0, len1);
			return y;
		}
		const uint32_t keep_tail = cw.tail, keep_n = cw.v.n;
		const wtz_aln_t y = wtz_align_zmer_lane_t(r1, len1, r2, len2, M, I, D, E, cw, true, &past);
		if(!past) return y;
		cw.tail = keep_tail; cw.v.n = keep_n;
	}
	return wtz_align_zmer_lane_t(pb1, len1, pb2, len2, M, I, D, E, cw, false, &past);
}

/* The fold chains window t (hzm_aln.h:1255-1300) over the results of its problems: init_score of a problem = the score so far, its
 * relative result is shifted by it - provided none of the absolute tests of kswx_extend_align_core would have fired (see wtz_sw_lane.h):
 * a + minrow > 0 (no row maximum <= 0: kswx.h:280,302) and, when the end candidate was taken, a + score > 0 (kswx.h:304).  A window that
 * fails is appended to fblist ([0] = count) for the chained kernel; nothing of it has been published. */
WTZ_HD void wtz_task_lfold(uint32_t t, const wtz_env_t &V, const wtz_wintask_t *tasks, const wtz_alnitem_t *items, const uint32_t *woff,
		const wtz_lprob_t *prob, const uint32_t *runoff, const uint32_t *runs, const wtz_lres_t *res, const uint8_t *wflag, uint32_t *fblist, const uint32_t *uidx, const uint32_t *nu){
	if(wflag[t]){ const uint32_t k = WTZ_ATOMIC_INC32(&fblist[0]); fblist[1 + k] = t; return; }
	const wtz_params_t *P = V.P;
	const wtz_alnitem_t &it = items[tasks[t].item];
	const wtz_win_t &w = it.win[tasks[t].widx];
	const wtz_readview pb1 = wtz_view(V.R, it.q, 0), pb2 = wtz_view(V.R, it.c, it.dir);
	const uint32_t base = woff[t], na = w.anchors[1] - w.anchors[0];
	const int32_t M = P->M, I = P->O, D = P->O, E = P->E;
	wtz_reg_t reg; memset(&reg, 0, sizeof reg);
	wtz_lcig_t cw; cw.v.init(V.pool, na * 14u + 16u); cw.tail = 0;
	wtz_aln_t x; memset(&x, 0, sizeof x);
	unsigned long long cells = 0; int32_t bad = 0;
	bool stop = false;
	const uint32_t n_used = nu[t];
	for(uint32_t u0 = 0; u0 < n_used && !stop; u0 += 2){
		/* the window's usable anchors (the planner's list) two at a time: slot index, then problem record / result / anchor / z-mer words -
		 * loads that do not depend on the chain are issued together, ahead of it */
		uint32_t KK[2]; wtz_lprob_t PR[2]; wtz_lres_t RES[2]; wtz_zhit_t AN[2]; uint32_t RO[2];
		#pragma unroll
		for(uint32_t j = 0; j < 2; j++) KK[j] = u0 + j < n_used ? uidx[base + u0 + j] : 0u;
		#pragma unroll
		for(uint32_t j = 0; j < 2; j++) if(u0 + j < n_used){ PR[j] = prob[base + KK[j]]; AN[j] = it.anchors[w.anchors[0] + KK[j]]; RES[j] = res[base + KK[j]]; RO[j] = runoff[base + KK[j]]; }
		#pragma unroll
		for(uint32_t j = 0; j < 2; j++){
			if(u0 + j >= n_used || stop) break;
			const wtz_lprob_t pr = PR[j];
			if(pr.qlen < 0) continue;
			const wtz_zhit_t p = AN[j];
			const int32_t off1 = (int32_t)ZH_OFF1(p), off2 = (int32_t)ZH_OFF2(p);
			if(x.aln == 0){ x.tb = x.te = off1; x.qb = x.qe = off2; }
			const int32_t a = x.score < 0 ? 0 : x.score;                 /* kswx.h:241 */
			if(pr.qlen > 0 && pr.tlen > 0){
				const wtz_lres_t r = RES[j];
				if(!(r.flags & WTZ_LR_DONE)){ bad = 1; stop = true; break; }
				if(a + r.minrow <= 0 || ((r.flags & WTZ_LR_USEDG) && a + r.score <= 0)){ const uint32_t z = WTZ_ATOMIC_INC32(&fblist[0]); fblist[1 + z] = t; return; }
				x.score = a + r.score;
				x.aln += r.mat + r.mis + r.ins + r.del; x.mat += r.mat; x.mis += r.mis; x.ins += r.ins; x.del += r.del;
				x.te += r.te; x.qe += r.qe;
				const uint32_t *rr = runs + RO[j];
				uint32_t jj = r.n_runs;
				for(; jj >= 4; jj -= 4){      /* batches of loads, then pushes */
					const uint32_t v0 = rr[jj - 1], v1 = rr[jj - 2], v2 = rr[jj - 3], v3 = rr[jj - 4];
					wtz_lcig_push(cw, v0 & 0xFu, v0 >> 4); wtz_lcig_push(cw, v1 & 0xFu, v1 >> 4); wtz_lcig_push(cw, v2 & 0xFu, v2 >> 4); wtz_lcig_push(cw, v3 & 0xFu, v3 >> 4);
				}
				for(; jj-- > 0;){ const uint32_t v = rr[jj]; wtz_lcig_push(cw, v & 0xFu, v >> 4); }
				cells += r.cells;
			} else x.score = a;                                          /* kswx.h:242: an empty side returns init_score */
			if(x.te < off1){ x.del += off1 - x.te; x.aln += off1 - x.te; wtz_lcig_push(cw, 2, (uint32_t)(off1 - x.te)); x.te = off1; }
			if(x.qe < off2){ x.ins += off2 - x.qe; x.aln += off2 - x.qe; wtz_lcig_push(cw, 1, (uint32_t)(off2 - x.qe)); x.qe = off2; }
			const uint32_t keep_tail = cw.tail, keep_n = cw.v.n;
			const wtz_aln_t y = wtz_align_zmer_lane(pb1.sub(off1, 1), ZH_LEN1(p), pb2.sub(off2, 1), ZH_LEN2(p), M, I, D, E, cw);
			if(y.aln == 0){ cw.tail = keep_tail; cw.v.n = keep_n; stop = true; break; }
			x.score += y.score;
			x.aln += y.aln; x.mat += y.mat; x.mis += y.mis; x.ins += y.ins; x.del += y.del;
			x.te += y.te; x.qe += y.qe;
		}
	}
	if(cw.tail){ cw.v.push(cw.tail); cw.tail = 0; }
	reg.x = x; reg.cigar = cw.v.a; reg.cigar_len = cw.v.n; reg.cells = cells;
	reg.pass = !(reg.x.aln * 2 < (int32_t)P->zovl || (float)reg.x.mat < (float)reg.x.aln * P->min_id);
	if(cw.v.bad || bad) reg.pass = 2;
	it.regs[tasks[t].widx] = reg;
}

#if defined(__HIPCC__)
/* four windows per wavefront (wtz_sw_grp.h): group g of the block takes window task t_base + g; a window outside the group form's
 * envelope is appended to `defer` ([0] = count) and redone by the one-window-per-wave kernel */
WTZ_D void wtz_task_winalign4(uint32_t t_base, uint32_t n, const wtz_env_t &V, const wtz_wintask_t *tasks, const wtz_alnitem_t *items, uint32_t *defer){
	const uint32_t g = threadIdx.x >> 4, gl = threadIdx.x & 15u;
	const uint32_t t = t_base + g;
	if(t >= n) return;
	const wtz_params_t *P = V.P;
	const wtz_alnitem_t &it = items[tasks[t].item];
	const wtz_win_t &w = it.win[tasks[t].widx];
	wtz_reg_t reg; memset(&reg, 0, sizeof reg);
	wtz_cigar_t cigar; cigar.init(V.pool, gl == 0 ? (w.anchors[1] - w.anchors[0]) * 14u + 16u : 0);
	unsigned long long cells = 0;
	bool ok = true, dfr = false;
	reg.x = wtz_align_window_grp(wtz_view(V.R, it.q, 0), wtz_view(V.R, it.c, it.dir), w, it.anchors, cigar, P, V.pool, (uint8_t*)wtz_wave_scratch() + (size_t)g * WTZ_GRP_LDS_BYTES, &cells, &ok, &dfr);
	if(gl != 0) return;
	if(dfr){ const uint32_t idx = atomicAdd(&defer[0], 1u); defer[1 + idx] = t; return; }
	reg.cigar = cigar.a; reg.cigar_len = cigar.n; reg.cells = cells;
	reg.pass = !(reg.x.aln * 2 < (int32_t)P->zovl || (float)reg.x.mat < (float)reg.x.aln * P->min_id);
	if(cigar.bad || !ok) reg.pass = 2;        /* pool exhausted */
	it.regs[tasks[t].widx] = reg;
}
#endif

/*
 * A10 (hzm_aln.h:1345-1486 with esti_regs = {0, len1}, wtzmo.c:1030) as three per-item phases around the two
 * K-sw3 extension jobs of an item, so that the extensions of a whole batch run as one wave-per-job launch:
 *   left  : pick the passing windows, describe the left extension (needs only the first window's score)
 *   mid   : fold the left result, fill the gaps between windows with K-sw2, describe the right extension
 *           (its init score is the running total, hzm_aln.h:1456)
 *   fin   : fold the right result, final kswx_t + CIGAR
 * The band-doubling loops around both extensions start at w = ew and stop at "w >= ew" (hzm_aln.h:1361-1374,
 * 1456-1468): exactly one call each.
 */
typedef struct { wtz_aln_t x; wtz_cigar_t cigar; uint32_t first, nreg; int32_t bad; unsigned long long cells_global;
	uint32_t mid;          /* 1: wtz_task_stitch_mid has run for this item (by the fused launch, wtz_kernel_stitch_ext_fr): K_stitch_mid skips it */
} wtz_stitch_state_t;

/* K-sw2 between two consecutive passing windows (hzm_aln.h:1386-1447), one task per window slot so that all gaps of a
 * batch run side by side; stored at the slot of the RIGHT window of the gap */
typedef struct { int32_t score, aln, mat, mis, ins, del; uint32_t *cigar; uint32_t cigar_len; int32_t bad, valid; unsigned long long cells; } wtz_gapres_t;
/* DP cells of ONE ksw_global2 call as its loops execute them: row i of the target covers query columns [max(i-w,0), min(i+w+1,qlen)) (ksw.c:529-533) */
WTZ_HD unsigned long long wtz_global_cells(int32_t qlen, int32_t tlen, int32_t w){
	unsigned long long n = 0;
	for(int32_t i = 0; i < tlen; i++){ const int32_t beg = i > w ? i - w : 0; int32_t end = i + w + 1; if(end > qlen) end = qlen; if(end > beg) n += (unsigned long long)(end - beg); }
	return n;
}

/* A gap whose band outgrows every register form (band doubling up to -W 3200 inside repeats: thousands of columns) used to run the
 * scalar body on lane 0 - seconds per gap.  The first launch (`wide_lds` = 0) now appends such a gap to `defer` ([0] = count) and a
 * second launch with WTZ_GAP_WIDE_LDS_BYTES of LDS per wave (`list` names its tasks) runs the LDS-ring wave DP with rings of 8192
 * columns and room for 32 k query bases; only what exceeds even that stays on the scalar body. */
#define WTZ_GAP_WIDE_LDS_BYTES (8192 + 2 * 8192 * 4)
typedef struct { int32_t score; bool from_reg, want_defer; int32_t bad; uint32_t *runs; uint32_t n_runs; int32_t r_mat, r_mis; int form; } wtz_gapdp_t;
#define WTZ_FORM_RING 32
#if defined(__HIP_DEVICE_COMPILE__)
/* ---- one K-sw2 problem (ksw_global2 at ONE band width w) on the wavefront: picks the device form from the shape - register DP with
 *      1 / 2 / 4 / 8 band columns per lane (4-bit trace in the LDS slice or in the pool), the LDS-ring wave DP (narrow slice, or the
 *      72 KB slice of the wide launch), the scalar body - and runs it.  TEST = true (wtz_test_dp, function-level parity vectors): `force`
 *      names the form (C | 16 = trace in the pool, 32 = LDS-ring wave DP, 255 = scalar body); R.form = -1 = outside that form's envelope. ---- */
template<bool TEST>
WTZ_D void wtz_gap_problem_wave(int32_t dq, const wtz_seq_packed &q, int32_t dt, const wtz_seq_packed &tt, const wtz_params_t *P, int32_t w,
		int32_t *lds, uint32_t wide_lds, bool may_defer, wtz_pool_t *pool, wtz_cigar_t &tmp, wtz_trace_t &tr, wtz_swmem_t &mem, int force, wtz_gapdp_t &R){
	const int32_t M = P->M, X = P->X, I = P->O, D = P->O, E = P->E;
	/* LDS slice: 128 sequence words (1 KB), then either the H/E rings of the general wave DP or the 4-bit trace of the
	 * register DP with its run list at the top end */
	wtz_wave_lds_t L; L.tb = (uint64_t*)lds; L.Hs = lds + 256; L.Es = lds + 768; L.PM = 511; L.tw = 128;
	wtz_wave_lds_t LW = L;           /* the wide launch: 1024 query words, then two rings of 8192 columns */
	if(wide_lds >= WTZ_GAP_WIDE_LDS_BYTES){ LW.Hs = lds + 2048; LW.Es = lds + 2048 + 8192; LW.PM = 8191; LW.tw = 1024; }
	uint8_t *ztr = (uint8_t*)(lds + 256); const int32_t ztr_bytes = WTZ_GAP_LDS_BYTES - 1024;
	R.score = 0; R.from_reg = false; R.want_defer = false; R.bad = 0; R.runs = NULL; R.n_runs = 0; R.r_mat = R.r_mis = 0; R.form = 0;
	int32_t score = 0;
	const int32_t n_col = dq < 2 * w + 1 ? dq : 2 * w + 1;
	const int32_t zrow = (n_col + 3) & ~3, run_bytes = 4 * (dq + dt + 4);
	const unsigned long long pt_g = WTZ_PROF_T(); (void)pt_g;
	const int32_t qwords = (dq + 63) / 32 + 1, qbytes = (qwords * 8 + 15) & ~15;
	bool lds_shape = (dq > 0 && dt > 0 && n_col <= 128 && dt <= 2048 && qwords <= 128 && ((dt + 1) / 2) * zrow + run_bytes <= ztr_bytes);
	bool reg_shape = lds_shape || (dq > 0 && dt > 0 && n_col <= 512 && qbytes + 4096 <= WTZ_GAP_LDS_BYTES);
	int32_t cmin = n_col <= 64 ? 1 : (n_col <= 128 ? 2 : (n_col <= 256 ? 4 : 8));
	bool ring_ok = true, scalar_only = false;
	if constexpr(TEST){
		if(force == 255){ lds_shape = reg_shape = false; ring_ok = false; scalar_only = true; may_defer = false; }
		else if(force == WTZ_FORM_RING){ lds_shape = reg_shape = false; may_defer = false; }
		else if(force){
			const int32_t fc = force & 15; const bool zg = (force & 16) != 0;
			const bool can = (fc == 1 || fc == 2 || fc == 4 || fc == 8) && fc >= cmin && n_col <= 64 * fc && (zg ? (dq > 0 && dt > 0 && qbytes + 4096 <= WTZ_GAP_LDS_BYTES) : (lds_shape && fc <= 2));
			if(!can){ R.form = -1; return; }
			lds_shape = !zg; reg_shape = true; cmin = fc;
		}
	}
	if(reg_shape && !lds_shape){
		/* the 4-bit trace does not fit the LDS slice (or the band is wider than 128 columns, or the gap longer than 2048 rows):
		 * it goes to the pool (HBM) and the traceback stages blocks of it in LDS.  Slice: query words | 4 KB stage | run list
		 * (when it still fits, else in the pool too) */
		uint8_t *stg = (uint8_t*)lds + qbytes;
		const bool runs_lds = (qbytes + 4096 + run_bytes <= WTZ_GAP_LDS_BYTES);
		unsigned long long za = 0, ra = 0;
		if(WTZ_LANE == 0){
			za = (unsigned long long)(uintptr_t)wtz_pool_alloc(pool, (size_t)((dt + 1) / 2) * zrow);
			if(!runs_lds) ra = (unsigned long long)(uintptr_t)wtz_pool_alloc(pool, (size_t)run_bytes);
		}
		za = __shfl(za, 0, 64); ra = __shfl(ra, 0, 64);
		if(za == 0 || (!runs_lds && ra == 0)){ R.bad = 1; score = 0; }
		else {
			R.runs = runs_lds ? (uint32_t*)(stg + 4096) : (uint32_t*)(uintptr_t)ra; R.from_reg = true; R.form = cmin | 16;
			uint8_t *zg = (uint8_t*)(uintptr_t)za;
			if(cmin == 1)      score = wtz_global_reg<1, true>(dq, q, dt, tt, M, X, -I, -E, -D, -E, w, L.tb, zg, (uint32_t)zrow, R.runs, &R.n_runs, &R.r_mat, &R.r_mis, stg);
			else if(cmin == 2) score = wtz_global_reg<2, true>(dq, q, dt, tt, M, X, -I, -E, -D, -E, w, L.tb, zg, (uint32_t)zrow, R.runs, &R.n_runs, &R.r_mat, &R.r_mis, stg);
			else if(cmin == 4) score = wtz_global_reg<4, true>(dq, q, dt, tt, M, X, -I, -E, -D, -E, w, L.tb, zg, (uint32_t)zrow, R.runs, &R.n_runs, &R.r_mat, &R.r_mis, stg);
			else               score = wtz_global_reg<8, true>(dq, q, dt, tt, M, X, -I, -E, -D, -E, w, L.tb, zg, (uint32_t)zrow, R.runs, &R.n_runs, &R.r_mat, &R.r_mis, stg);
		}
		WTZ_PROF_ADD(43, pt_g); WTZ_PROF_CNT(44, 1000000); WTZ_PROF_MAX(45, pt_g);
	} else if(lds_shape){
		R.runs = (uint32_t*)(ztr + ztr_bytes - run_bytes); R.from_reg = true; R.form = cmin;
		if(cmin == 1) score = wtz_global_reg<1>(dq, q, dt, tt, M, X, -I, -E, -D, -E, w, L.tb, ztr, (uint32_t)zrow, R.runs, &R.n_runs, &R.r_mat, &R.r_mis);
		else          score = wtz_global_reg<2>(dq, q, dt, tt, M, X, -I, -E, -D, -E, w, L.tb, ztr, (uint32_t)zrow, R.runs, &R.n_runs, &R.r_mat, &R.r_mis);
		WTZ_PROF_ADD(32, pt_g); WTZ_PROF_CNT(33, 1000000); WTZ_PROF_MAX(34, pt_g);
	} else if(may_defer && dq > 0 && dt > 0 && !(n_col + 2 <= 512 && qwords <= 128) && n_col + 2 <= 8192 && qwords <= 1024 && (dt + 63) / 64 <= WTZ_TRACE_MAXCHUNK){
		R.want_defer = true;       /* uniform: the wide launch redoes this gap */
		return;
	} else if(ring_ok && dq > 0 && dt > 0 && n_col + 2 <= LW.PM + 1 && qwords <= LW.tw && (dt + 63) / 64 <= WTZ_TRACE_MAXCHUNK){
		bool ok = true;
		R.form = WTZ_FORM_RING;
		score = wtz_global_wave(dq, q, dt, tt, M, X, -I, -E, -D, -E, w, LW, tr, pool, tmp, &ok);
		if(!ok) R.bad = 1;
#ifdef WTZ_GAP_CHECK
		if(ok && WTZ_LANE == 0){
			wtz_cigar_t t2; t2.init(pool, 32); wtz_swmem_t m2; wtz_swmem_init(m2, pool);
			const int32_t s2 = wtz_global_banded(dq, q, dt, tt, M, X, -I, -E, -D, -E, w, m2, t2);
			uint32_t firstdiff = 0xFFFFFFFFu;
			if(!m2.bad){ if(t2.n != tmp.n) firstdiff = 0xFFFFFFFEu; else for(uint32_t z9 = 0; z9 < t2.n; z9++) if(t2.a[z9] != tmp.a[z9]){ firstdiff = z9; break; } }
			if(!m2.bad && (s2 != score || firstdiff != 0xFFFFFFFFu)) printf("[gap-check] dq %d dt %d w %d n_col %d: score wave %d scalar %d; cigar n %u / %u first diff %u\n", dq, dt, w, n_col, score, s2, tmp.n, t2.n, firstdiff);
		}
#endif
		WTZ_PROF_ADD(35, pt_g); WTZ_PROF_CNT(36, 1000000); WTZ_PROF_MAX(37, pt_g);
		WTZ_PROF_CNT(46, (unsigned long long)dt * 1000); if(n_col > 256) WTZ_PROF_CNT(47, 1000000);
	} else {
		if constexpr(TEST){ if(force == WTZ_FORM_RING){ R.form = -1; return; } }
		(void)scalar_only;
		score = 0; R.form = 255;
		if(WTZ_LANE == 0){ score = wtz_global_banded(dq, q, dt, tt, M, X, -I, -E, -D, -E, w, mem, tmp); if(mem.bad) R.bad = 1; }
		WTZ_PROF_ADD(38, pt_g); WTZ_PROF_CNT(39, 1000000); WTZ_PROF_MAX(40, pt_g);
	}
	R.score = __shfl(score, 0, 64);
	R.bad = __shfl(R.bad, 0, 64) ? 1 : 0;
}
#endif

WTZ_HD void wtz_task_gap(uint32_t t, const wtz_env_t &V, const wtz_wintask_t *tasks, const wtz_alnitem_t *items, wtz_gapres_t *gaps,
		uint32_t *defer = NULL, const uint32_t *list = NULL, uint32_t wide_lds = 0){
	if(list) t = list[1 + t];
	const wtz_params_t *P = V.P;
	const wtz_alnitem_t &it = items[tasks[t].item];
	const uint32_t k = tasks[t].widx;
	wtz_gapres_t g; memset(&g, 0, sizeof g);
	const unsigned long long pt_task = WTZ_PROF_T(); (void)pt_task;
	wtz_gapres_t *slot = gaps + (it.regs - items[0].regs) + k;        /* regs of all items are one contiguous array */
	if(k == 0 || it.regs[k].pass != 1){ if(WTZ_LANE == 0) *slot = g; return; }
	int32_t prev = -1;
	for(int32_t j = (int32_t)k - 1; j >= 0; j--) if(it.regs[j].pass == 1){ prev = j; break; }
	if(prev < 0){ if(WTZ_LANE == 0) *slot = g; return; }
	const int32_t M = P->M, X = P->X, I = P->O, D = P->O, E = P->E;
	const wtz_reg_t *reg1 = &it.regs[prev], *reg2 = &it.regs[k];
	const wtz_readview pb1 = wtz_view(V.R, it.q, 0), pb2 = wtz_view(V.R, it.c, it.dir);
	const int32_t dq = reg2->x.qb - reg1->x.qe, dt = reg2->x.tb - reg1->x.te;
	const wtz_seq_packed q = pb2.sub(reg1->x.qe, 1), tt = pb1.sub(reg1->x.te, 1);
	wtz_cigar_t tmp; tmp.init(V.pool, WTZ_LANE == 0 ? 32 : 0);
	int32_t w = P->w, score, bad = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	bool from_reg = false; uint32_t n_runs = 0; uint32_t *runs = NULL; int32_t r_mat = 0, r_mis = 0;
	{   /* the whole wavefront computes the banded global alignment; band doubling is uniform (score is broadcast) */
		(void)M; (void)X; (void)I; (void)D; (void)E;
		int32_t *lds = wtz_wave_scratch();
		wtz_trace_t tr; tr.chunk = NULL; tr.zb = NULL; tr.n_chunk = 0; tr.zrow = 0; tr.cap_rows = 0;
		wtz_swmem_t mem; wtz_swmem_init(mem, V.pool);
		for(;;){
			if(w < WTZ_ABSDIFF(dq, dt)){ w <<= 1; continue; }
			wtz_gapdp_t G;
			wtz_gap_problem_wave<false>(dq, q, dt, tt, P, w, lds, wide_lds, defer != NULL, V.pool, tmp, tr, mem, 0, G);
			if(G.want_defer){
				if(WTZ_LANE == 0){ const uint32_t idx = atomicAdd(&defer[0], 1u); defer[1 + idx] = t; }
				return;
			}
			score = G.score; from_reg = G.from_reg; runs = G.runs; n_runs = G.n_runs; r_mat = G.r_mat; r_mis = G.r_mis;
			if(WTZ_LANE == 0) g.cells += wtz_global_cells(dq, dt, w);         /* every call of the band-doubling loop counts (hzm_aln.h:1400-1417) */
			if(G.bad){ bad = 1; break; }
			if(score < 0 && w < P->W && w < WTZ_MAX(dq, dt)) w <<= 1;
			else break;
		}
	}
	if(WTZ_LANE != 0) return;
	if(from_reg){     /* runs are in traceback order; match / mismatch counts came with them */
		g.score = score; g.valid = 1; g.mat = r_mat; g.mis = r_mis;
		tmp.n = 0;
		for(uint32_t k = n_runs; k-- > 0;){
			const uint32_t r = runs[k], op = r & 0xFu; const int32_t len = (int32_t)(r >> 4);
			tmp.push(r);
			g.aln += len;
			if(op == 1) g.ins += len; else if(op == 2) g.del += len;
		}
		g.cigar = tmp.a; g.cigar_len = tmp.n; g.bad = (tmp.bad || bad);
		*slot = g;
		WTZ_PROF_ADD(41, pt_task); WTZ_PROF_MAX(42, pt_task);
		return;
	}
#else
	{
		wtz_swmem_t mem; wtz_swmem_init_lds(mem, V.pool, wtz_wave_scratch(), WTZ_WAVE_LDS_BYTES / 4);
		for(;;){
			if(w < WTZ_ABSDIFF(dq, dt)){ w <<= 1; continue; }
			score = wtz_global_banded(dq, q, dt, tt, M, X, -I, -E, -D, -E, w, mem, tmp);
			g.cells += wtz_global_cells(dq, dt, w);
			if(score < 0 && w < P->W && w < WTZ_MAX(dq, dt)) w <<= 1;
			else break;
		}
		if(mem.bad) bad = 1;
	}
#endif
	g.score = score; g.valid = 1;
	int32_t x1 = 0, x2 = 0;
	for(uint32_t idx = 0; idx < tmp.n; idx++){
		int32_t op = (int32_t)(tmp.a[idx] & 0xF), len = (int32_t)(tmp.a[idx] >> 4);
		g.aln += len;
		if(op == 0){ for(int32_t j = 0; j < len; j++){ if(q.at(x1 + j) == tt.at(x2 + j)) g.mat++; else g.mis++; } x1 += len; x2 += len; }
		else if(op == 1){ x1 += len; g.ins += len; }
		else if(op == 2){ x2 += len; g.del += len; }
	}
	g.cigar = tmp.a; g.cigar_len = tmp.n; g.bad = (tmp.bad || bad);
	*slot = g;
}

/* ---------------- K-sw2 with one lane per gap (wtz_lane_global) ----------------
 * The gaps between the passing windows of an item are independent problems (hzm_aln.h:1386-1447).  Slot t of the window list holds the gap in
 * front of window t (wtz_task_gap): the planner lists the gaps whose FIRST band width - P->w doubled until it covers |dq - dt| - fits the lane
 * envelope; the DP kernel runs them one per lane, sorted by shape, and publishes a gap unless the reference would double the band again
 * (score < 0, hzm_aln.h:1410-1416); whatever is not published (done[t] == 0) goes through wtz_task_gap on a wavefront. */
#define WTZ_LG_MAXROWS 768       /* rows of a lane gap: longer gaps are few and long, a wavefront each serves them better */
typedef struct { uint32_t slot; int32_t dq, dt, w; } wtz_lgap_t;
WTZ_HD void wtz_task_gplan(uint32_t t, const wtz_env_t &V, const wtz_wintask_t *tasks, const wtz_alnitem_t *items, wtz_gapres_t *gaps,
		wtz_lgap_t *gp, uint32_t *runcap, uint64_t *key, uint32_t *val, uint8_t *done, uint32_t *ccnt){
	const wtz_params_t *P = V.P;
	const wtz_alnitem_t &it = items[tasks[t].item];
	const uint32_t k = tasks[t].widx;
	wtz_lgap_t G; G.slot = t; G.dq = G.dt = 0; G.w = 0;
	uint32_t cap = 0; uint64_t ky = 0x3FFFFull; uint8_t dn = 0; int32_t cls = -1;
	wtz_gapres_t *slot = gaps + (it.regs - items[0].regs) + k;
	int32_t prev = -1;
	if(k && it.regs[k].pass == 1) for(int32_t j = (int32_t)k - 1; j >= 0; j--) if(it.regs[j].pass == 1){ prev = j; break; }
	if(prev < 0){ wtz_gapres_t g; memset(&g, 0, sizeof g); *slot = g; dn = 1; }       /* no gap in front of this window (wtz_task_gap's early exits) */
	else {
		const wtz_reg_t *reg1 = &it.regs[prev], *reg2 = &it.regs[k];
		const int32_t dq = reg2->x.qb - reg1->x.qe, dt = reg2->x.tb - reg1->x.te;
		if(dq > 0 && dt > 0){
			int32_t w = P->w; while(w < WTZ_ABSDIFF(dq, dt)) w <<= 1;
			const int32_t n_col = dq < 2 * w + 1 ? dq : 2 * w + 1;
			G.dq = dq; G.dt = dt; G.w = w;      /* also for the gaps the lanes do not take: the wavefront kernel's launch is ordered by them (run_gap_lane) */
			if(n_col <= WTZ_LN_MAXCOLS && dt <= WTZ_LG_MAXROWS){
				cap = (uint32_t)(dq + dt + 2);
				ky = 0x3FFFFull - (uint64_t)(((uint32_t)n_col << 11) | (uint32_t)dt);
				cls = (int32_t)wtz_lane_class(n_col);
			}
		}
	}
	/* the four class counters: one atomic per class and wavefront (round 6: every lane bumped one of four addresses - ~10 M same-address atomics per configs[2] step in a
	 * kernel that does nothing else but a few dependent loads) */
#if defined(__HIP_DEVICE_COMPILE__)
	#pragma unroll
	for(int32_t c4 = 0; c4 < 4; c4++){
		const unsigned long long m = __ballot(cls == c4);
		if(m && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m)) atomicAdd(&ccnt[c4], (uint32_t)__popcll(m));
	}
#else
	if(cls >= 0) WTZ_ATOMIC_INC32(&ccnt[cls]);
#endif
	gp[t] = G; runcap[t] = cap; key[t] = ky; val[t] = t; done[t] = dn;
}
template<int NC>
WTZ_HD void wtz_task_gdp(uint32_t gw, uint32_t wv, const wtz_env_t &V, const wtz_wintask_t *tasks, const wtz_alnitem_t *items, const uint32_t *order, uint32_t lo, uint32_t hi,
		const wtz_lgap_t *gp, wtz_lres_t *res, uint64_t *wtr, uint32_t *wrm){
	const wtz_params_t *P = V.P;
	const uint32_t idx = lo + wv * WTZ_NLANES + WTZ_LANE;
	const bool live = idx < hi;
	const uint32_t t = live ? order[idx] : 0u;
	wtz_lgap_t G; G.slot = 0; G.dq = G.dt = 0; G.w = P->w;
	if(live) G = gp[t];
	const int32_t M = P->M, X = P->X, I = P->O, D = P->O, E = P->E;
	wtz_seq_packed q, tt; q.bits = tt.bits = V.R.bits; q.start = tt.start = 0; q.strand = tt.strand = 1; q.comp = tt.comp = 0;
	if(live){
		const wtz_alnitem_t &it = items[tasks[t].item];
		const uint32_t k = tasks[t].widx;
		int32_t prev = -1;
		for(int32_t j = (int32_t)k - 1; j >= 0; j--) if(it.regs[j].pass == 1){ prev = j; break; }
		const wtz_reg_t *reg1 = &it.regs[prev];
		q = wtz_view(V.R, it.c, it.dir).sub(reg1->x.qe, 1); tt = wtz_view(V.R, it.q, 0).sub(reg1->x.te, 1);
	}
	constexpr uint32_t RS = (uint32_t)wtz_lane_geo<NC>::RS;
	const uint32_t rows_max = (uint32_t)wtz_lane_wmax(G.dt);
	uint64_t pa = 0;
	if(WTZ_LANE == 0){
		pa = (uint64_t)(uintptr_t)wtz_pool_alloc(V.pool + 1, (size_t)WTZ_NLANES * rows_max * RS * 4 + 16);
		wtr[gw] = pa; wrm[gw] = rows_max;
	}
	pa = wtz_coop_bcast64(pa);
	if(pa == 0) return;                      /* nothing published: the wave kernel takes these gaps (and reports the pool if it is really full) */
	uint32_t *tr = (uint32_t*)(uintptr_t)pa;      /* the wave's block; the lanes' rows are interleaved inside it (wtz_ltr_at) */
	wtz_lres_t R; memset(&R, 0, sizeof R);
	wtz_lane_global<NC>(live, G.dq, q, G.dt, tt, G.w, M, X, -I, -E, -D, -E, tr, R);
	if(live) res[t] = R;
}
/* traceback + publication of the gaps of the same wave */
WTZ_HD void wtz_task_gtb(uint32_t gw, uint32_t wv, uint32_t RS, const wtz_env_t &V, const wtz_wintask_t *tasks, const wtz_alnitem_t *items, const uint32_t *order, uint32_t lo, uint32_t hi,
		const wtz_lgap_t *gp, const uint32_t *runoff, uint32_t *runs, const wtz_lres_t *res, wtz_gapres_t *gaps, uint8_t *done, const uint64_t *wtr, const uint32_t *wrm){
	const wtz_params_t *P = V.P;
	const uint32_t idx = lo + wv * WTZ_NLANES + WTZ_LANE;
	const uint64_t pa = wtr[gw];
	if(idx >= hi || pa == 0) return;
	const uint32_t t = order[idx];
	const wtz_lgap_t G = gp[t];
	wtz_lres_t R = res[t];
	if(!(R.flags & WTZ_LR_DONE)) return;
	if(R.score < 0 && G.w < (int32_t)P->W && G.w < WTZ_MAX(G.dq, G.dt)) return;      /* the reference doubles the band and tries again (hzm_aln.h:1410-1416): wave kernel */
	const wtz_alnitem_t &it = items[tasks[t].item];
	const uint32_t k = tasks[t].widx;
	int32_t prev = -1;
	for(int32_t j = (int32_t)k - 1; j >= 0; j--) if(it.regs[j].pass == 1){ prev = j; break; }
	const wtz_reg_t *reg1 = &it.regs[prev];
	const wtz_seq_packed q = wtz_view(V.R, it.c, it.dir).sub(reg1->x.qe, 1), tt = wtz_view(V.R, it.q, 0).sub(reg1->x.te, 1);
	const uint32_t *tr = (const uint32_t*)(uintptr_t)pa;      /* lane-interleaved, as the DP kernel wrote it */
	uint32_t *rr = runs + runoff[t];
	const int32_t r0 = G.dt - 1, c0 = (r0 + G.w + 1 < G.dq ? r0 + G.w + 1 : G.dq) - 1;
	wtz_lane_traceback<true>(true, r0, c0, G.w, RS, tt, G.dt, q, G.dq, tr, rr, R);
	wtz_gapres_t g; memset(&g, 0, sizeof g);
	g.score = R.score; g.valid = 1; g.mat = R.mat; g.mis = R.mis; g.ins = R.ins; g.del = R.del; g.aln = R.mat + R.mis + R.ins + R.del;
	for(uint32_t a = 0, b = R.n_runs; a + 1 < b; a++, b--){ const uint32_t x = rr[a]; rr[a] = rr[b - 1]; rr[b - 1] = x; }      /* traceback order -> CIGAR order, in place */
	g.cigar = rr; g.cigar_len = R.n_runs; g.cells = R.cells;
	gaps[(it.regs - items[0].regs) + k] = g; done[t] = 1;
}
WTZ_HD void wtz_task_gdp_all(uint32_t wv, const wtz_lclass_t &L, const wtz_env_t &V, const wtz_wintask_t *tasks, const wtz_alnitem_t *items, const uint32_t *order,
		const wtz_lgap_t *gp, wtz_lres_t *res, uint64_t *wtr, uint32_t *wrm){
	uint32_t k = 0; while(k < 3 && wv >= L.wend[k]) k++;
	const uint32_t wl = wv - (k ? L.wend[k - 1] : 0u);
	if(k == 0)      wtz_task_gdp<104>(wv, wl, V, tasks, items, order, L.lo[0], L.hi[0], gp, res, wtr, wrm);
	else if(k == 1) wtz_task_gdp<64>(wv, wl, V, tasks, items, order, L.lo[1], L.hi[1], gp, res, wtr, wrm);
	else if(k == 2) wtz_task_gdp<32>(wv, wl, V, tasks, items, order, L.lo[2], L.hi[2], gp, res, wtr, wrm);
	else            wtz_task_gdp<16>(wv, wl, V, tasks, items, order, L.lo[3], L.hi[3], gp, res, wtr, wrm);
}
WTZ_HD void wtz_task_gtb_all(uint32_t wv, const wtz_lclass_t &L, const wtz_env_t &V, const wtz_wintask_t *tasks, const wtz_alnitem_t *items, const uint32_t *order,
		const wtz_lgap_t *gp, const uint32_t *runoff, uint32_t *runs, const wtz_lres_t *res, wtz_gapres_t *gaps, uint8_t *done, const uint64_t *wtr, const uint32_t *wrm){
	uint32_t k = 0; while(k < 3 && wv >= L.wend[k]) k++;
	const uint32_t wl = wv - (k ? L.wend[k - 1] : 0u);
	wtz_task_gtb(wv, wl, 16u >> k, V, tasks, items, order, L.lo[k], L.hi[k], gp, runoff, runs, res, gaps, done, wtr, wrm);
}
WTZ_HD void wtz_task_glist(uint32_t t, const uint8_t *done, uint32_t *list){ if(!done[t]){ const uint32_t k = WTZ_ATOMIC_INC32(&list[0]); list[1 + k] = t; } }

/* rgeo (optional): qlen / tlen of the RIGHT extension wtz_task_stitch_mid will ask for (-1: none) - they follow from the last window that passed alone, so the
 * launch that runs both extensions of an item on one wavefront can order its items and plan its trace memory before any extension has run */
WTZ_HD void wtz_task_stitch_left(uint32_t t, const wtz_env_t &V, const wtz_alnitem_t *items, wtz_stitch_state_t *sts, wtz_extjob_t *jobs, int32_t *rgeo = NULL){
	const wtz_params_t *P = V.P;
	const wtz_alnitem_t &it = items[t];
	wtz_stitch_state_t st; memset(&st, 0, sizeof st);
	wtz_extjob_t jb; memset(&jb, 0, sizeof jb); jb.item = t;
	st.first = 0xFFFFFFFFu;
	for(uint32_t k = 0; k < it.nwin; k++){ if(it.regs[k].pass == 2) st.bad = 1; if(it.regs[k].pass == 1){ if(st.first == 0xFFFFFFFFu) st.first = k; st.nreg++; } }
	if(st.nreg && !st.bad){
		st.cigar.init(V.pool, 256);
		st.x = it.regs[st.first].x;
		if(st.x.qb && st.x.tb){
			const wtz_readview pb1 = wtz_view(V.R, it.q, 0), pb2 = wtz_view(V.R, it.c, it.dir);
			jb.valid = 1; jb.qlen = st.x.qb; jb.tlen = st.x.tb; jb.q = pb2.sub(st.x.qb - 1, -1); jb.t = pb1.sub(st.x.tb - 1, -1);
			jb.init_score = st.x.score + 100 * P->M; jb.W = -P->ew;
		}
	}
	if(rgeo){
		int32_t rq = -1, rt = -1;
		if(st.nreg && !st.bad){
			int32_t qe = st.x.qe, te = st.x.te;
			for(uint32_t k = st.first + 1; k < it.nwin; k++) if(it.regs[k].pass == 1){ qe = it.regs[k].x.qe; te = it.regs[k].x.te; }
			const int32_t len1 = (int32_t)V.R.rdlen[it.q], len2 = (int32_t)V.R.rdlen[it.c];
			if(te < len1 && qe < len2){ rq = len2 - qe; rt = len1 - te; }
		}
		rgeo[2 * (size_t)t] = rq; rgeo[2 * (size_t)t + 1] = rt;
	}
	sts[t] = st; jobs[t] = jb;
}

WTZ_HD void wtz_task_stitch_mid(uint32_t t, const wtz_env_t &V, const wtz_alnitem_t *items, wtz_stitch_state_t *sts, const wtz_extjob_t *jobsL, wtz_extjob_t *jobsR, const wtz_gapres_t *gaps, bool fused = false){
	const wtz_params_t *P = V.P;
	const wtz_alnitem_t &it = items[t];
	const int32_t M = P->M;
	const bool l0 = (WTZ_LANE == 0);
	wtz_stitch_state_t st = sts[t];
	if(st.mid) return;          /* done by the fused launch */
	wtz_extjob_t jr; memset(&jr, 0, sizeof jr); jr.item = t;
	if(st.nreg == 0 || st.bad){ if(l0){ jobsR[t] = jr; if(fused) sts[t].mid = 1; } return; }
	const wtz_readview pb1 = wtz_view(V.R, it.q, 0), pb2 = wtz_view(V.R, it.c, it.dir);
	const int32_t len1 = (int32_t)pb1.len, len2 = (int32_t)pb2.len;
	const wtz_gapres_t *gp = gaps + (it.regs - items[0].regs);
	wtz_aln_t x = st.x;
	if(jobsL[t].valid){
		const wtz_extjob_t &jl = jobsL[t];
		if(jl.bad) st.bad = 1;
		const wtz_aln_t y = jl.x;
		x.score = y.score - 100 * M;
		x.aln += y.aln; x.mat += y.mat; x.mis += y.mis; x.ins += y.ins; x.del += y.del;
		x.qb -= y.qe; x.tb -= y.te;
	}
	/* the operation lists in order: left extension (back to front), first window, then gap + window for every later window that passed */
	const uint32_t first = st.first, nwr = it.nwin - first - 1;
	const wtz_reg_t *regs = it.regs;
	const uint32_t *lc = jobsL[t].valid ? jobsL[t].cigar : NULL; const uint32_t ln = jobsL[t].valid ? jobsL[t].cigar_len : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
	uint32_t *jlds = (uint32_t*)wtz_wave_scratch(); const uint32_t jwords = WTZ_WAVE_LDS_BYTES / 4;
#else
	std::vector<uint64_t> jbuf(3 * (size_t)(2u + 2u * nwr) + 1); uint32_t *jlds = (uint32_t*)jbuf.data(); const uint32_t jwords = (uint32_t)(jbuf.size() * 2);
#endif
#ifdef WTZ_NO_CIGAR_JOIN
	const bool joined = false; (void)jlds; (void)jwords; (void)lc;
#else
	const bool joined = wtz_cigar_join_coop(st.cigar, 2u + 2u * nwr, [=](uint32_t p, const uint32_t **src, uint32_t *n, uint32_t *rev){
		*rev = 0;
		if(p == 0){ *src = lc; *n = ln; *rev = 1; return; }
		if(p == 1){ *src = regs[first].cigar; *n = regs[first].cigar_len; return; }
		const uint32_t k = first + 1 + ((p - 2) >> 1);
		if(regs[k].pass != 1){ *src = NULL; *n = 0; return; }
		if(p & 1u){ *src = regs[k].cigar; *n = regs[k].cigar_len; } else { *src = gp[k].cigar; *n = gp[k].cigar_len; }
	}, jlds, jwords);
#endif
	if(!joined){
		if(ln){ wtz_cigar_reverse_coop(jobsL[t].cigar, ln); wtz_cigar_concat_coop(st.cigar, jobsL[t].cigar, ln); }
		wtz_cigar_concat_coop(st.cigar, regs[first].cigar, regs[first].cigar_len);
	}
	for(uint32_t k = first + 1; k < it.nwin; k++){
		if(regs[k].pass != 1) continue;
		const wtz_reg_t *reg2 = &regs[k];
		const wtz_gapres_t &g = gp[k];
		if(!g.valid || g.bad) st.bad = 1;
		st.cells_global += g.cells;
		x.score += g.score;
		x.aln += g.aln; x.mat += g.mat; x.mis += g.mis; x.ins += g.ins; x.del += g.del;
		if(!joined) wtz_cigar_concat_coop(st.cigar, g.cigar, g.cigar_len);
		x.score += reg2->x.score;
		x.aln += reg2->x.aln; x.mat += reg2->x.mat; x.mis += reg2->x.mis; x.ins += reg2->x.ins; x.del += reg2->x.del;
		x.qe = reg2->x.qe; x.te = reg2->x.te;
		if(!joined) wtz_cigar_concat_coop(st.cigar, reg2->cigar, reg2->cigar_len);
	}
	if(st.cigar.bad) st.bad = 1;
	if(x.te < len1 && x.qe < len2){
		jr.valid = 1; jr.qlen = len2 - x.qe; jr.tlen = len1 - x.te; jr.q = pb2.sub(x.qe, 1); jr.t = pb1.sub(x.te, 1);
		jr.init_score = x.score; jr.W = -P->ew;
	}
	if(l0){ st.x = x; st.mid = fused ? 1u : 0u; sts[t] = st; jobsR[t] = jr; }
}

WTZ_HD void wtz_task_stitch_fin(uint32_t t, const wtz_env_t &V, const wtz_alnitem_t *items, wtz_stitch_state_t *sts, const wtz_extjob_t *jobsL, const wtz_extjob_t *jobsR, wtz_alnres_dev_t *out){
	wtz_stitch_state_t st = sts[t];
	wtz_alnres_dev_t r; memset(&r, 0, sizeof r);
	r.n_regs = st.nreg; r.bad = st.bad; r.cells_global = st.cells_global;
	for(uint32_t k = 0; k < items[t].nwin; k++) r.cells_fixed += items[t].regs[k].cells;
	if(st.nreg && !st.bad){
		wtz_aln_t x = st.x;
		if(jobsR[t].valid){
			const wtz_extjob_t &jr = jobsR[t];
			if(jr.bad) r.bad = 1;
			const wtz_aln_t y = jr.x;
			x.score = y.score;
			x.aln += y.aln; x.mat += y.mat; x.mis += y.mis; x.ins += y.ins; x.del += y.del;
			x.qe += y.qe; x.te += y.te;
			wtz_cigar_concat_coop(st.cigar, jr.cigar, jr.cigar_len);
			r.cells_shift += jr.cells;
		}
		if(jobsL[t].valid) r.cells_shift += jobsL[t].cells;
		if(st.cigar.bad) r.bad = 1;
		r.x = x; r.cigar = st.cigar.a; r.cigar_len = st.cigar.n; r.text_len = wtz_cigar_text_len_coop(st.cigar.a, st.cigar.n);
	}
	if(WTZ_LANE == 0) out[t] = r;
}

/* A11 (-n): re-align the stitched overlap of item t inside the band drawn around its own CIGAR (wtzmo.c:1031-1034) and replace
 * result + CIGAR.  Wave-cooperative on the GPU; the host emulation has no wave and runs the plain loops. */
#define WTZ_REFINE_LDS_BYTES (2048 * 4 * 2 + 1032 * 8)
#if !defined(__HIP_DEVICE_COMPILE__)
static inline bool wtz_refine_scalar(const wtz_seq_packed &query, int32_t qb, const wtz_seq_packed &target, int32_t tb, int32_t W, int32_t M, int32_t X, int32_t I, int32_t D, int32_t E,
		const uint32_t *cig, uint32_t ncig, wtz_pool_t *pool, wtz_cigar_t &out, wtz_aln_t *res){
	wtz_aln_t y; memset(&y, 0, sizeof y);
	out.n = 0;
	int32_t qe = qb, te = tb;
	for(uint32_t i = 0; i < ncig; i++){ const uint32_t op = cig[i] & 0xFu; const int32_t len = (int32_t)(cig[i] >> 4); if(op == 0){ qe += len; te += len; } else if(op == 1) qe += len; else te += len; }
	const int32_t ql = qe - qb, tl = te - tb;
	if(ql == 0 || tl == 0){ *res = y; return true; }
	int32_t *zw = (int32_t*)wtz_pool_alloc(pool, (size_t)(ql + 2) * (4 * 3 + 8) + (size_t)(tl + 2) * 8);
	if(zw == NULL) return false;
	/* the 64-bit array leads the (16-byte aligned) block so that it is 8-byte aligned for every ql */
	unsigned long long *zoff = (unsigned long long*)zw; zw = (int32_t*)(zoff + (ql + 2));
	int32_t *zb = zw + (ql + 2), *ze = zb + (ql + 2);
	int32_t *rh = ze + (ql + 2), *re = rh + (tl + 2);
	for(int32_t i = 0; i < ql + 2; i++) zw[i] = 0;
	int32_t qx = 0, tx = 0;
	for(uint32_t i = 0; i < ncig; i++){
		const uint32_t op = cig[i] & 0xFu; const int32_t len = (int32_t)(cig[i] >> 4);
		if(op == 0){ for(int32_t j = 0; j < len; j++) zw[qx++] = W; } else if(op == 1){ for(int32_t j = 0; j < len; j++) zw[qx++] = W + len; }
	}
	qx = 0;
	for(uint32_t i = 0; i < ncig; i++){
		const uint32_t op = cig[i] & 0xFu; const int32_t len = (int32_t)(cig[i] >> 4);
		if(op == 0) qx += len;
		else if(op == 1){ for(int32_t j = 1; j < len && j < qx; j++) zw[qx - j] += len - j; qx += len - 1; for(int32_t j = 1; j < len && j + qx < ql; j++) zw[qx + j] += len - j; qx++; }
		else { for(int32_t j = 1; j < len && j < qx; j++) zw[qx - j] += len - j; for(int32_t j = 1; j < len && j + qx < ql; j++) zw[qx + j] += len - j; }
	}
	qx = 0;
	for(uint32_t i = 0; i < ncig; i++){
		const uint32_t op = cig[i] & 0xFu; const int32_t len = (int32_t)(cig[i] >> 4);
		if(op == 0 || op == 1){ for(int32_t j = 0; j < len; j++){ int32_t b = tx - zw[qx]; if(b < 0) b = 0; int32_t e = tx + 1 + zw[qx]; if(e > tl) e = tl; zb[qx] = b; ze[qx] = e; if(op == 0) tx++; qx++; } }
		else tx += len;
	}
	{ int32_t b = 0; for(int32_t i = 0; i < ql; i++){ if(zb[i] < b) zb[i] = b; else if(zb[i] > b) b = zb[i]; } }
	{ int32_t e = tl; for(int32_t i = ql - 1; i >= 0; i--){ if(ze[i] > e) ze[i] = e; else if(ze[i] < e) e = ze[i]; } }
	unsigned long long ztot = 0;
	for(int32_t i = 0; i < ql; i++){ zoff[i] = ztot; ztot += (unsigned long long)(ze[i] > zb[i] ? ze[i] - zb[i] : 0); }
	uint8_t *z = (uint8_t*)wtz_pool_alloc(pool, (size_t)ztot + 64);
	if(z == NULL) return false;
	rh[0] = 0; for(int32_t j = 1; j <= tl; j++) rh[j] = -10000;
	for(int32_t j = 0; j <= tl; j++) re[j] = -10000;
	for(int32_t i = 0; i < ql; i++){
		const uint32_t qc = query.at(i);
		int32_t h1 = -10000, f = -10000, j;
		uint8_t *zi = z + zoff[i];
		for(j = zb[i]; j < ze[i]; j++){
			const bool eq = (qc == target.at(j));
			int32_t m = rh[j] + (eq ? M : X), e, h, t; uint32_t d;
			rh[j] = h1;
			e = re[j];
			if(m >= e){ d = 0; h = m; } else { d = 1; h = e; }
			if(h < f){ d = 2; h = f; }
			h1 = h;
			t = m + I + E; e = e + E; if(e > t) d |= 1u << 2; else e = t;
			re[j] = e;
			t = m + D + E; f = f + E; if(f > t) d |= 2u << 4; else f = t;
			if(eq) d |= 0x80u;
			zi[j - zb[i]] = (uint8_t)d;
		}
		rh[j] = h1; re[j] = -10000;
	}
	y.qb = qb; y.qe = qe; y.tb = tb; y.te = te; y.score = rh[tl];
	int32_t i_ = ql - 1, j_ = tl - 1; uint32_t d_ = 0;
	while(i_ >= 0 && j_ >= 0){
		const uint32_t zv = (j_ >= zb[i_] && j_ < ze[i_]) ? z[zoff[i_] + (unsigned long long)(j_ - zb[i_])] : 0u;
		d_ = (zv >> (d_ << 1)) & 0x03;
		if(d_ == 0){ if(zv & 0x80u) y.mat++; else y.mis++; i_--; j_--; }
		else if(d_ == 1){ i_--; y.ins++; }
		else { j_--; y.del++; }
		wtz_cigar_push(out, d_, 1);
	}
	if(i_ >= 0){ y.ins += i_ + 1; wtz_cigar_push(out, 1, (uint32_t)(i_ + 1)); }
	if(j_ >= 0){ y.del += j_ + 1; wtz_cigar_push(out, 2, (uint32_t)(j_ + 1)); }
	wtz_cigar_reverse(out.a, out.n);
	y.aln = y.mat + y.mis + y.ins + y.del;
	*res = y;
	return true;
}
#endif

WTZ_HD void wtz_task_refine(uint32_t t, const wtz_env_t &V, const wtz_alnitem_t *items, wtz_alnres_dev_t *res){
	const wtz_params_t *P = V.P;
	wtz_alnres_dev_t r = res[t];
	if(r.n_regs == 0 || r.bad) return;
	const wtz_alnitem_t &it = items[t];
	if(P->aux_strand){
		/* align_hzmaux gates the STITCHED alignment before it refines it (hzm_aln.h:1715-1718; the refined one is returned unchecked, 1721-1729):
		 * an item that fails leaves as "no regions" (n_regs 0), the caller's sign for "no hit" */
		const int32_t tl = (int32_t)V.R.rdlen[it.q], ql = (int32_t)V.R.rdlen[it.c];
		int32_t beg = r.x.qb - r.x.tb; if(beg < 0) beg = 0;
		int32_t end = r.x.qe + tl - r.x.te; if(end > ql) end = ql;
		const int32_t ovl = end - beg;
		if(r.x.score < 0 || (float)r.x.mat < (float)r.x.aln * P->min_id || (float)r.x.mat < (float)ovl * P->min_id){
			if(WTZ_LANE == 0){ r.n_regs = 0; res[t] = r; }
			return;
		}
	}
	const wtz_readview pb1 = wtz_view(V.R, it.q, 0), pb2 = wtz_view(V.R, it.c, it.dir);
	wtz_cigar_t out; out.a = NULL; out.n = out.cap = 0; out.pool = V.pool; out.bad = 0;
	if(WTZ_LANE == 0) out.init(V.pool, r.cigar_len + 64);
	wtz_aln_t y; memset(&y, 0, sizeof y);
	bool ok;
#if defined(__HIP_DEVICE_COMPILE__)
	int32_t *lds = wtz_wave_scratch();
	wtz_wave_lds_t L; L.Hs = lds; L.Es = lds + 2048; L.tb = (uint64_t*)(lds + 4096); L.PM = 2047; L.tw = 1032;
	ok = wtz_refine_wave(pb2.sub(r.x.qb, 1), r.x.qb, pb1.sub(r.x.tb, 1), r.x.tb, P->w, P->M, P->X, P->O, P->O, P->E, r.cigar, r.cigar_len, L, V.pool, out, &y);
	if(WTZ_LANE != 0) return;
#else
	ok = wtz_refine_scalar(pb2.sub(r.x.qb, 1), r.x.qb, pb1.sub(r.x.tb, 1), r.x.tb, P->w, P->M, P->X, P->O, P->O, P->E, r.cigar, r.cigar_len, V.pool, out, &y);
#endif
	if(!ok || out.bad){ r.bad = 1; res[t] = r; return; }
	r.x = y; r.cigar = out.a; r.cigar_len = out.n; r.text_len = wtz_cigar_text_len(out.a, out.n);
	res[t] = r;
}

/* scalar execution of one extension job: host emulation, over-size jobs, and the on-device cross-check */
WTZ_HD void wtz_task_extjob_scalar(uint32_t t, const wtz_env_t &V, wtz_extjob_t *jobs){
	const wtz_params_t *P = V.P;
	wtz_extjob_t &jb = jobs[t];
	if(!jb.valid) return;
	wtz_swmem_t mem; wtz_swmem_init(mem, V.pool);
	wtz_cigar_t cg; cg.init(V.pool, 64);
	unsigned long long cells = 0;
	jb.x = wtz_extend_shift(jb.qlen, jb.q, jb.tlen, jb.t, jb.init_score, jb.W, P->M, P->X, P->O, P->O, P->E, P->T, mem, cg, &cells);
	jb.cigar = cg.a; jb.cigar_len = cg.n; jb.bad = (mem.bad || cg.bad); jb.cells = cells;
}

#endif
