/*
 * wtz_sw_lane.h — the fixed-band DPs with ONE LANE PER PROBLEM (64 independent problems per wavefront):
 *     K-sw1  kswx_extend_align_core   kswx.h:234-335   (wtz_lane_fixed)
 *     K-sw2  ksw_global2              ksw.c:503-586    (wtz_lane_global)
 *
 * Why lanes and not wavefronts.  The K-sw1 problems of a step (configs[2]: 85 M of them) are tiny — 41 % have no side longer than 16
 * bases, 82 % none longer than 64 — so a wavefront per problem leaves most of its 64 lanes idle on every row and spends as many
 * instructions on the one-lane sections (traceback, CIGAR) as on cells: ≈7 600 wave instructions per problem.  What made the wave form
 * necessary was the chain inside a window (hzm_aln.h:1247-1302): problem k+1 starts where problem k ended, with problem k's score as its
 * `init_score`.  But
 *   (1) the START of every problem is known without running any DP: after the gap fill and the patch of hzm_aln.h:1273-1284 the cursor
 *       stands exactly on the anchor, and the z-mer run alignment (hzm_aln.h:278-314) moves it by amounts that depend on the two
 *       sequences only.  A planner that walks the anchors (wtz_task_lplan) therefore lists all problems of all windows up front;
 *   (2) the DP itself is translation invariant in init_score except for three tests against absolute constants: the row maximum is
 *       clamped at 0 and a row whose maximum is <= 0 ends the loop (kswx.h:280,302), the end candidate starts at 0 (kswx.h:271,300-301,
 *       304), and the band clamp max_gap (kswx.h:244-247).  So every problem is run with init_score = 0 ("relative mode", REL) with
 *       those tests left open, and records what the fold needs to decide them once the real init_score is known: the smallest row
 *       maximum and whether the end candidate was used.  The fold (wtz_task_lfold, one lane per window) chains the scores; a window in
 *       which a test would have gone the other way (0.3 % of the problems, all at init_score < 60) is redone by the exact chained kernel
 *       (wtz_task_winalign), as is any window with a problem outside the envelope below.
 * The -10000 sentinels are absolute too, but a real cell value is >= init_score - 5 (rows + columns) - 8: with rows + columns <= 1500
 * no real value ever comes near them in either mode, so they lose every comparison in both (checked by the planner per problem).
 *
 * Registers, not LDS.  A lane keeps H and E of its whole band in VGPRs: RH[c] = H(i-1, jb+c-1) (the reference's rh[] is one column
 * off, kswx.h:276-277), RE[c] = E(i, jb+c), band-relative; the band moves right by S = 0 or 1 columns per row (S = 1 once i > W), so
 * cell c reads slot c+S and writes slot c — ascending in place, no copies.  The target bases of the band are a register window of
 * 2-bit codes that shifts in one base per S = 1 row; the query base of a row comes from a 64-bit word refilled every 32 rows.  All
 * indices are compile-time constants (the column loop is fully unrolled, NC = 16 / 32 / 64 / 104 columns per class), problems are sorted
 * by shape so that the lanes of a wavefront run the same trip counts, and blocks of 8 columns beyond the widest band of the wave are
 * skipped.  Trace: 4 bits per cell - the sign bits of the four differences the recurrence decides on (m - E, max(m, E) - F, open - extend for E and for F),
 * pushed with one v_alignbit_b32 each, no compare / select pair - in the transient pool, the rows of a wavefront's 64 lanes
 * INTERLEAVED (wtz_ltr_at): a row is one to four 16-byte stores per lane, and a store instruction of the wave covers consecutive bytes.
 *
 * The same bodies compile for the host emulation (tests/emul: one lane per "wavefront"), so the planner / fold logic and the DP are
 * exercised by the CPU test-suite against the reference goldens.
 */
#ifndef WTZ_SW_LANE_H
#define WTZ_SW_LANE_H

#include "wtz_sw.h"
#include "wtz_sw_wave.h"

#define WTZ_LN_MAXCOLS 104          /* band columns of the widest class: 2 * 51 + 1 >= n_col of the default -w 50 */
#define WTZ_LN_MAXROWS 511          /* rows of a problem (9 bits of the sort key) */
#define WTZ_LN_MAXSPAN 1500         /* rows + columns: keeps every real cell value far above the -10000 sentinels */
#define WTZ_LN_NEG (-10000)
#define WTZ_LN_LOW (-(1 << 29))     /* REL: "no value yet" of the row maximum / the end candidate */

/* acc = acc << 1 | sign(x): one v_alignbit_b32; eight cells push four sign bits each into one trace dword */
#if defined(__HIP_DEVICE_COMPILE__)
WTZ_D uint32_t wtz_push_sign(uint32_t acc, int32_t x){ return __builtin_amdgcn_alignbit(acc, (uint32_t)x, 31u); }
#else
WTZ_COOP_HOST uint32_t wtz_push_sign(uint32_t acc, int32_t x){ return (acc << 1) | ((uint32_t)x >> 31); }
#endif
#define WTZ_LN_OFF (-(1 << 28))     /* diagonal value of a cell beyond the end of the lane's band: loses every comparison, overflows nothing */
/* maximum over the lanes of the wavefront, uniform (the host emulation has one lane) */
#if defined(__HIP_DEVICE_COMPILE__)
WTZ_D int32_t wtz_lane_wmax(int32_t v){ return wtz_wave_max_i32(v); }      /* DPP reduction + v_readlane: the result is an SGPR, branches on it are scalar */
#else
WTZ_COOP_HOST int32_t wtz_lane_wmax(int32_t v){ return v; }
#endif

/* one problem slot per anchor of a window (hzm_aln.h:1262-1272): where the gap before the anchor starts and how long it is */
typedef struct {
	uint32_t win;                 /* window task */
	int32_t qoff, toff;           /* x.qe / x.te before the problem (candidate axis / query-read axis) */
	int32_t qlen, tlen;           /* off2 - x.qe, off1 - x.te; qlen < 0: the anchor is skipped (hzm_aln.h:1262-1263) or lies behind the end of the window's walk */
	uint32_t run_off;             /* first entry of the problem's run list */
} wtz_lprob_t;
#define WTZ_LR_USEDG 1u
#define WTZ_LR_DONE 2u
typedef struct {
	int32_t score;                /* REL: relative to init_score */
	int32_t qe, te;               /* kswx_t: one past the end cell */
	int32_t mat, mis, ins, del;
	int32_t minrow;               /* REL: smallest row maximum (relative) */
	uint32_t n_runs, flags, cells;
} wtz_lres_t;

template<int NC> struct wtz_lane_geo {
	static constexpr int KW = (NC + 15) / 16;                                    /* 32-bit words of the target window */
	static constexpr int RS = NC <= 16 ? 2 : (NC <= 32 ? 4 : (NC <= 64 ? 8 : 16));  /* trace dwords per row */
};
WTZ_HD uint32_t wtz_lane_rs(int32_t n_col){ return n_col <= 16 ? 2u : (n_col <= 32 ? 4u : (n_col <= 64 ? 8u : 16u)); }
WTZ_HD uint32_t wtz_lane_class(int32_t n_col){ return n_col <= 16 ? 0u : (n_col <= 32 ? 1u : (n_col <= 64 ? 2u : 3u)); }
/* Trace layout of a wavefront (round 4): LANE-INTERLEAVED.  A row of a lane is RS dwords, stored in units of U = min(RS, 4) dwords (one 8- or 16-byte store);
 * unit q of row i of lane l sits at dword ((i * RS / U + q) * WTZ_NLANES + l) * U of the wave's block, i.e. the 64 lanes' units of one (row, q) are
 * contiguous: a store instruction of the DP writes 512 / 1024 consecutive bytes instead of 64 scattered 8 / 16-byte pieces (K_ldp wrote 246 GB per
 * configs[2] step for 84 GB of trace: partial lines), and the walkers of a wave - sorted by shape, so they move through their rows together - read
 * neighbouring dwords of the same lines instead of one 64-byte line per lane and step (K_ltb fetched 448 GB per step).  With one lane per
 * "wavefront" (host emulation) the formula is the old per-lane row-major layout. */
WTZ_HD size_t wtz_ltr_at(uint32_t row, uint32_t RS, uint32_t d){
	const uint32_t lu = RS >= 4u ? 2u : 1u;                    /* log2 U */
	return (((((size_t)row * (RS >> lu)) + (d >> lu)) * WTZ_NLANES + WTZ_LANE) << lu) + (d & ((1u << lu) - 1u));
}

/* register-tail run writer of one lane: the open run stays in a register (kswx_push_cigar merges equal neighbours, kswx.h:39-44) */
typedef struct { uint32_t *a; uint32_t n, tail; } wtz_lruns_t;
WTZ_HD void wtz_lruns_push(wtz_lruns_t &w, uint32_t op, uint32_t len){
	if(len == 0) return;
	if(w.tail && (w.tail & 0xFu) == op) w.tail += len << 4;
	else { if(w.tail) w.a[w.n++] = w.tail; w.tail = (len << 4) | op; }
}
WTZ_HD void wtz_lruns_finish(wtz_lruns_t &w){ if(w.tail){ w.a[w.n++] = w.tail; w.tail = 0; } }

/*
 * K-sw1 of one lane.  `live` = this lane has a problem; every lane of the wavefront must call (uniform trip counts come from wave-wide
 * maxima).  ABS = true: the reference's function for a known init_score (function-level tests, and the form that could chain);
 * ABS = false: relative mode, see the header.  W / ql / tl: wtz_ext_geometry of the problem.  tr: this lane's trace rows (ql rows of
 * RS dwords; the WAVE's block, lane-interleaved: wtz_ltr_at), runs: room for ql + tl + 2 runs, written in TRACEBACK order (the reader reverses).
 */
template<int NC, bool ABS>
WTZ_HD void wtz_lane_fixed(bool live, int32_t qlen, const wtz_seq_packed &q, int32_t tlen, const wtz_seq_packed &t, int32_t init_score,
		int32_t W, int32_t ql, int32_t tl, int32_t M, int32_t X, int32_t I, int32_t D, int32_t E, int32_t T, uint32_t *tr, wtz_lres_t &R){
	constexpr int KW = wtz_lane_geo<NC>::KW, RS = wtz_lane_geo<NC>::RS;
	const int32_t a = ABS ? (init_score < 0 ? 0 : init_score) : 0;
	int32_t RH[NC + 1], RE[NC + 1]; uint32_t TW[KW];
	#pragma unroll
	for(int c = 0; c <= NC; c++){ RH[c] = c == 0 ? a : a + D + E * c; RE[c] = WTZ_LN_NEG; }       /* kswx.h:266-268 */
	#pragma unroll
	for(int k = 0; k < KW; k += 2){
		const uint64_t w = live ? wtz_pack32(t, 16 * k, tlen) : 0ull;
		TW[k] = (uint32_t)w; if(k + 1 < KW) TW[k + 1] = (uint32_t)(w >> 32);
	}
	uint64_t tfeed = 0, qw = 0; int32_t tnext = 16 * KW;        /* next target base to shift into the window */
	int32_t best = a, bi = -1, bj = -1, gmax = ABS ? 0 : WTZ_LN_LOW, gi = -1, gj = -1, minrow = 1 << 29;
	bool stopped = !live; uint32_t ncell = 0;
	const int32_t rows_max = wtz_lane_wmax(live ? ql : 0);
	for(int32_t i = 0; i < rows_max; i++){
		const bool on = !stopped && i < ql;
		const int32_t jb = i > W ? i - W : 0, je = i + W + 1 < tl ? i + W + 1 : tl;
		const int32_t n = on ? je - jb : 0;
		const bool S = i > W;                                   /* the band start moved by one column against the previous row */
		if(on && S){
			#pragma unroll
			for(int k = 0; k + 1 < KW; k++) TW[k] = (TW[k] >> 2) | (TW[k + 1] << 30);
			const int32_t ph = (tnext - 16 * KW) & 31;
			if(ph == 0) tfeed = wtz_pack32(t, tnext, tlen);
			TW[KW - 1] = (TW[KW - 1] >> 2) | ((uint32_t)((tfeed >> (2 * ph)) & 3ull) << 30);
			tnext++;
		}
		if((i & 31) == 0) qw = on ? wtz_pack32(q, i, qlen) : 0ull;
		const uint32_t qb = (uint32_t)(qw >> (2 * (i & 31))) & 3u;
		uint32_t NE[KW];                                        /* bit 2c of word c / 16: target base jb + c differs from the row's query base */
		#pragma unroll
		for(int k = 0; k < KW; k++){ const uint32_t x = TW[k] ^ (qb * 0x55555555u); NE[k] = (x | (x >> 1)) & 0x55555555u; }
		const int32_t XM = X - M;
		/* kswx.h:274-297, branch-free: a cell at or beyond the end of the lane's band (c >= n) computes too but only hands h1 on and closes
		 * its slots with (h1, <= -10000) - slot n is rh[je] = h1; re[je] = -10000 (kswx.h:298), the slots behind it are never read again
		 * before they are rewritten (a row reads up to the closing slot of the row above).  Row maximum and its LAST arg-max (kswx.h:288-289)
		 * come from one running maximum of (h << 7) + c + 1: the start value 0 is "imax = 0, mj2 = -1", a negative h never beats it. */
		int32_t h1 = jb == 0 ? a + I + E * (i + 1) : WTZ_LN_NEG, f = WTZ_LN_NEG;
		int32_t key = ABS ? 0 : (int32_t)0x80000000;
		const int32_t nmax = wtz_lane_wmax(n);
		uint32_t acc[4] = {0u, 0u, 0u, 0u};
		uint32_t *trow = tr + wtz_ltr_at((uint32_t)i, (uint32_t)RS, 0u);      /* unit 0 of the row; unit q is q * 4 * WTZ_NLANES dwords further */
		#pragma unroll
		for(int c0 = 0; c0 < NC; c0 += 8){
			if(c0 <= nmax){                                   /* uniform: some lane of the wave has a cell (or its closing slot) in this block of 8 columns */
				#pragma unroll
				for(int c = c0; c < c0 + 8 && c < NC; c++){
					const bool act = c < n;
					int32_t hd = S ? RH[c + 1] : RH[c]; const int32_t ev = S ? RE[c + 1] : RE[c];
					hd = act ? hd : WTZ_LN_OFF;                         /* beyond the band: m, h and the new E all end up below every real value */
					const int32_t ne = (int32_t)((NE[c >> 4] >> (2 * (c & 15))) & 1u);     /* 1: the bases differ */
					const int32_t m = hd + M + ne * XM;
					RH[c] = h1;
					uint32_t a4 = acc[(c >> 3) & 3];
					a4 = wtz_push_sign(a4, m - ev);                     /* bit 3 of the cell's nibble: m < E (the vertical gap wins, kswx.h:281) */
					int32_t h = m >= ev ? m : ev;
					a4 = wtz_push_sign(a4, h - f);                      /* bit 2: max(m, E) < F (the horizontal gap wins, kswx.h:283) */
					h = h >= f ? h : f;
					h1 = act ? h : h1;
					const int32_t kc = (int32_t)(((uint32_t)h << 7) | (uint32_t)(c + 1));
					key = key > kc ? key : kc;                          /* a cell beyond the band carries h = F < an earlier cell's h: it never wins */
					int32_t tt = m + I + E; const int32_t e2 = ev + E;
					a4 = wtz_push_sign(a4, tt - e2);                    /* bit 1: E extended (kswx.h:291) */
					RE[c] = e2 > tt ? e2 : tt;                          /* beyond the band: <= -10000 - 1, the closing slot's sentinel (it loses every comparison like -10000 itself) */
					tt = m + D + E; const int32_t f2 = f + E;
					a4 = wtz_push_sign(a4, tt - f2);                    /* bit 0: F extended (kswx.h:295) */
					f = f2 > tt ? f2 : tt;
					acc[(c >> 3) & 3] = a4;
				}
			}
			if(((c0 + 8) & 31) == 0 || c0 + 8 >= NC){          /* a group of 32 columns is complete: one 16-byte store */
				if(on && (c0 & ~31) < n){
					if(RS == 2){ trow[0] = acc[0]; trow[1] = acc[1]; }
					else { uint32_t *p = trow + (size_t)(c0 >> 5) * 4u * WTZ_NLANES; p[0] = acc[0]; p[1] = acc[1]; p[2] = acc[2]; p[3] = acc[3]; }
				}
				acc[0] = acc[1] = acc[2] = acc[3] = 0u;
			}
		}
		if(n == NC){ RH[NC] = h1; RE[NC] = WTZ_LN_NEG; }
		const int32_t imax = key >> 7, mj2 = (key & 127) ? jb + (key & 127) - 1 : -1;
		if(on){
			ncell += (uint32_t)n;
			if(je == tlen && gmax < h1){ gmax = h1; gi = i; gj = je - 1; }          /* kswx.h:299-301 */
			if(i + 1 == qlen && gmax < imax){ gmax = imax; gi = i; gj = mj2; }
			if(imax > best){ best = imax; bi = i; bj = mj2; }
			else if(ABS && imax <= 0) stopped = true;                               /* kswx.h:302 */
			if(!ABS && imax < minrow) minrow = imax;
		}
	}
	const bool useg = ABS ? (gmax > 0 && gmax >= best + T) : (gmax > WTZ_LN_LOW && gmax >= best + T);      /* kswx.h:304-308 */
	const int32_t i_ = useg ? gi : bi, j_ = useg ? gj : bj;
	R.score = useg ? gmax : best; R.qe = i_ + 1; R.te = j_ + 1;
	R.minrow = minrow; R.flags = (useg ? WTZ_LR_USEDG : 0u) | WTZ_LR_DONE; R.cells = ncell;
}

/* The walk back through a lane's trace rows (kswx.h:309-333 / ksw.c:560-586 as wtz_trace_walk states them), a function of its own: it is a
 * chain of dependent loads - one trace dword per step - and nothing else, so the product runs it in a kernel of its own with four times
 * the resident waves of the register-heavy DP kernels (inside them it took as long as the cells: two waves per SIMD cannot hide it).
 * GLOBAL = false: K-sw1 from the end cell (r, c) = (R.qe - 1, R.te - 1), row-only steps are I (1), column-only steps D (2);
 * GLOBAL = true: K-sw2 from the corner, rows are the TARGET: row-only steps are D, column-only steps I, `rseq` is the target.
 * rseq / rlen = the sequence along the rows, cseq / clen = along the band columns.  runs: traceback order. */
template<bool GLOBAL>
WTZ_HD void wtz_lane_traceback(bool live, int32_t r, int32_t cc, int32_t W, uint32_t RS, const wtz_seq_packed &rseq, int32_t rlen, const wtz_seq_packed &cseq, int32_t clen,
		const uint32_t *tr, uint32_t *runs, wtz_lres_t &R){
	int32_t mat = 0, mis = 0, rowonly = 0, colonly = 0, rblk = -1, cblk = -1; uint64_t rw = 0, cw = 0;
	uint32_t state = 0;
	wtz_lruns_t Wr; Wr.a = runs; Wr.n = 0; Wr.tail = 0;
	if(!live){ r = -1; cc = -1; }
	while(r >= 0 && cc >= 0){
		const int32_t jb = r > W ? r - W : 0, c = cc - jb;
		const uint32_t nib = (tr[wtz_ltr_at((uint32_t)r, RS, (uint32_t)(c >> 3))] >> (4 * (7 - (c & 7)))) & 15u;      /* cell 0 of a block sits in the top nibble; bits: 3 m < E, 2 max(m, E) < F, 1 E extended, 0 F extended */
		state = state == 0 ? ((nib & 4u) ? 2u : (nib >> 3)) : (state == 1 ? ((nib & 2u) ? 1u : 0u) : ((nib & 1u) ? 2u : 0u));
		if((r >> 5) != rblk){ rblk = r >> 5; rw = wtz_pack32(rseq, rblk * 32, rlen); }
		if((cc >> 5) != cblk){ cblk = cc >> 5; cw = wtz_pack32(cseq, cblk * 32, clen); }
		if(state == 0){
			if(((rw >> (2 * (r & 31))) & 3ull) == ((cw >> (2 * (cc & 31))) & 3ull)) mat++; else mis++;
			r--; cc--; wtz_lruns_push(Wr, 0, 1);
		} else if(state == 1){ r--; rowonly++; wtz_lruns_push(Wr, GLOBAL ? 2u : 1u, 1); }
		else { cc--; colonly++; wtz_lruns_push(Wr, GLOBAL ? 1u : 2u, 1); }
	}
	if(live && r >= 0){ rowonly += r + 1; wtz_lruns_push(Wr, GLOBAL ? 2u : 1u, (uint32_t)(r + 1)); }
	if(live && cc >= 0){ colonly += cc + 1; wtz_lruns_push(Wr, GLOBAL ? 1u : 2u, (uint32_t)(cc + 1)); }
	wtz_lruns_finish(Wr);
	R.mat = mat; R.mis = mis; R.ins = GLOBAL ? colonly : rowonly; R.del = GLOBAL ? rowonly : colonly; R.n_runs = Wr.n;
}

/*
 * K-sw2 of one lane: ksw_global2 (ksw.c:503-586) at ONE band width w, the way wtz_global_banded (wtz_sw.h) states it: rows over the TARGET,
 * band columns over the query, -0x40000000 outside the band, no early exit, score = corner cell, the walk starts there.  Same register
 * scheme as wtz_lane_fixed: RH[c] = feed[jb + c] (H of the previous row, one column off), RE[c] = vertical gap state of column jb + c.
 * Requires |qlen - tlen| <= w (the caller's band doubling, hzm_aln.h:1400-1417, starts there) so that no row is empty.
 * runs: traceback order, ops already as CIGAR ops (0 M, 1 I = query only, 2 D = target only).
 */
template<int NC>
WTZ_HD void wtz_lane_global(bool live, int32_t qlen, const wtz_seq_packed &q, int32_t tlen, const wtz_seq_packed &t, int32_t w,
		int32_t M, int32_t X, int32_t o_del, int32_t e_del, int32_t o_ins, int32_t e_ins, uint32_t *tr, wtz_lres_t &R){
	constexpr int KW = wtz_lane_geo<NC>::KW, RS = wtz_lane_geo<NC>::RS;
	const int32_t MINF = WTZ_MINUS_INF, open_v = o_del + e_del, open_g = o_ins + e_ins;
	int32_t RH[NC + 1], RE[NC + 1]; uint32_t QW[KW];
	#pragma unroll
	for(int c = 0; c <= NC; c++){ RH[c] = c == 0 ? 0 : (c <= w ? -(o_ins + e_ins * c) : MINF); RE[c] = MINF; }
	#pragma unroll
	for(int k = 0; k < KW; k += 2){
		const uint64_t v = live ? wtz_pack32(q, 16 * k, qlen) : 0ull;
		QW[k] = (uint32_t)v; if(k + 1 < KW) QW[k + 1] = (uint32_t)(v >> 32);
	}
	uint64_t qfeed = 0, tw = 0; int32_t qnext = 16 * KW;
	(void)RS;
	int32_t left_last = 0, je_last = 0; uint32_t ncell = 0;
	const int32_t rows_max = wtz_lane_wmax(live ? tlen : 0);
	for(int32_t i = 0; i < rows_max; i++){
		const bool on = live && i < tlen;
		const int32_t jb = i > w ? i - w : 0, je = i + w + 1 < qlen ? i + w + 1 : qlen;
		const int32_t n = on ? je - jb : 0;
		const bool S = i > w;
		if(on && S){
			#pragma unroll
			for(int k = 0; k + 1 < KW; k++) QW[k] = (QW[k] >> 2) | (QW[k + 1] << 30);
			const int32_t ph = (qnext - 16 * KW) & 31;
			if(ph == 0) qfeed = wtz_pack32(q, qnext, qlen);
			QW[KW - 1] = (QW[KW - 1] >> 2) | ((uint32_t)((qfeed >> (2 * ph)) & 3ull) << 30);
			qnext++;
		}
		if((i & 31) == 0) tw = on ? wtz_pack32(t, i, tlen) : 0ull;
		const uint32_t tb = (uint32_t)(tw >> (2 * (i & 31))) & 3u;
		uint32_t NE[KW];
		#pragma unroll
		for(int k = 0; k < KW; k++){ const uint32_t x = QW[k] ^ (tb * 0x55555555u); NE[k] = (x | (x >> 1)) & 0x55555555u; }
		const int32_t XM = X - M;
		int32_t left = jb == 0 ? -(o_del + e_del * (i + 1)) : MINF, g = MINF;
		const int32_t nmax = wtz_lane_wmax(n);
		uint32_t acc[4] = {0u, 0u, 0u, 0u};
		uint32_t *trow = tr + wtz_ltr_at((uint32_t)i, (uint32_t)RS, 0u);      /* unit 0 of the row; unit q is q * 4 * WTZ_NLANES dwords further */
		#pragma unroll
		for(int c0 = 0; c0 < NC; c0 += 8){
			if(c0 <= nmax){
				#pragma unroll
				for(int c = c0; c < c0 + 8 && c < NC; c++){        /* branch-free like wtz_lane_fixed: cells at or beyond the band end hand `left` on and close their slots */
					const bool act = c < n;
					int32_t hd = S ? RH[c + 1] : RH[c]; const int32_t v = S ? RE[c + 1] : RE[c];
					hd = act ? hd : MINF;
					const int32_t ne = (int32_t)((NE[c >> 4] >> (2 * (c & 15))) & 1u);
					const int32_t m = hd + M + ne * XM;
					uint32_t a4 = acc[(c >> 3) & 3];
					a4 = wtz_push_sign(a4, m - v);
					int32_t h = m >= v ? m : v;
					a4 = wtz_push_sign(a4, h - g);
					h = h >= g ? h : g;
					RH[c] = left; left = act ? h : left;
					const int32_t vo = m - open_v, ve = v - e_del;
					a4 = wtz_push_sign(a4, vo - ve);
					RE[c] = ve > vo ? ve : vo;
					const int32_t go = m - open_g, ge = g - e_ins;
					a4 = wtz_push_sign(a4, go - ge);
					g = ge > go ? ge : go;
					acc[(c >> 3) & 3] = a4;
				}
			}
			if(((c0 + 8) & 31) == 0 || c0 + 8 >= NC){
				if(on && (c0 & ~31) < n){
					if(RS == 2){ trow[0] = acc[0]; trow[1] = acc[1]; }
					else { uint32_t *p = trow + (size_t)(c0 >> 5) * 4u * WTZ_NLANES; p[0] = acc[0]; p[1] = acc[1]; p[2] = acc[2]; p[3] = acc[3]; }
				}
				acc[0] = acc[1] = acc[2] = acc[3] = 0u;
			}
		}
		if(n == NC){ RH[NC] = left; RE[NC] = MINF; }
		if(on){ ncell += (uint32_t)n; left_last = left; je_last = je; }
	}
	/* feed[qlen]: the corner cell when the last row reaches the last column, else what row -1 left there */
	R.score = je_last == qlen ? left_last : (qlen == 0 ? 0 : (qlen <= w ? -(o_ins + e_ins * qlen) : MINF));
	R.cells = ncell; R.flags = WTZ_LR_DONE; R.minrow = 0;
	R.qe = qlen; R.te = tlen;
}

/* up to 64 bases of a view in two registers (base i at bits 2i); bases at or beyond the length given to wtz_pack32 read as 0 */
struct wtz_seq_w2 { uint64_t w0, w1;
	WTZ_HDM uint32_t at(int32_t i) const { return (uint32_t)(((i < 32) ? (w0 >> (2 * i)) : (w1 >> (2 * (i - 32)))) & 3u); } };
WTZ_HD wtz_seq_w2 wtz_seq_load_w2(const wtz_seq_packed &s, uint32_t len){
	wtz_seq_w2 r; r.w0 = wtz_pack32(s, 0, (int32_t)len); r.w1 = len > 32 ? wtz_pack32(s, 32, (int32_t)len) : 0ull; return r;
}

/* hzm_aln.h:278-314 without the CIGAR: where the z-mer run alignment leaves the cursor (te, qe), and whether the pair aligns at all.
 * The two expanded z-mers are loaded once as 2-bit words (a base-by-base walk is a chain of dependent HBM loads, and the lanes of a wave
 * reach their usable anchors at different iterations: the planner spent 8 of its 9 ms there); the run loop leaves the registers only when
 * it would read past either z-mer (the reference then compares with whatever follows in the read: wtz_zmer_advance_slow does the same). */
template<typename S1, typename S2>
WTZ_HD bool wtz_zmer_advance_slow(const S1 &pb1, uint32_t len1, const S2 &pb2, uint32_t len2, int32_t *dte, int32_t *dqe){
	uint32_t s0 = 0, s1 = 0;
	while(s0 < len1 || s1 < len2){
		const uint32_t b = pb1.at((int32_t)s0);
		if(b != pb2.at((int32_t)s1)) return false;
		uint32_t e0 = s0 + 1; while(e0 < len1 && pb1.at((int32_t)e0) == b) e0++;
		uint32_t e1 = s1 + 1; while(e1 < len2 && pb2.at((int32_t)e1) == b) e1++;
		s0 = e0; s1 = e1;
	}
	*dte = (int32_t)s0; *dqe = (int32_t)s1;      /* te = mat + del = sum of l0, qe = mat + ins = sum of l1 */
	return s0 != 0 || s1 != 0;                   /* aln == 0 (two empty z-mers) ends the window like a mismatch (hzm_aln.h:1288-1291) */
}
/* on the two z-mers already in registers (len1, len2 <= 64); *past = true: the loop would read beyond one of them (nothing decided) */
WTZ_HD bool wtz_zmer_advance_w2(const wtz_seq_w2 &r1, uint32_t len1, const wtz_seq_w2 &r2, uint32_t len2, int32_t *dte, int32_t *dqe, bool *past){
	if(len1 == len2 && r1.w0 == r2.w0 && r1.w1 == r2.w1){ *dte = (int32_t)len1; *dqe = (int32_t)len2; return len1 != 0; }      /* the same string: every run pairs in full */
	uint32_t s0 = 0, s1 = 0;
	while(s0 < len1 || s1 < len2){
		if(s0 >= len1 || s1 >= len2){ *past = true; return false; }
		const uint32_t b = r1.at((int32_t)s0);
		if(b != r2.at((int32_t)s1)) return false;
		uint32_t e0 = s0 + 1; while(e0 < len1 && r1.at((int32_t)e0) == b) e0++;
		uint32_t e1 = s1 + 1; while(e1 < len2 && r2.at((int32_t)e1) == b) e1++;
		s0 = e0; s1 = e1;
	}
	*dte = (int32_t)s0; *dqe = (int32_t)s1;
	return true;
}
WTZ_HD bool wtz_zmer_advance(const wtz_seq_packed &pb1, uint32_t len1, const wtz_seq_packed &pb2, uint32_t len2, int32_t *dte, int32_t *dqe){
	if(len1 <= 64 && len2 <= 64){
		bool past = false;
		const bool ok = wtz_zmer_advance_w2(wtz_seq_load_w2(pb1, len1), len1, wtz_seq_load_w2(pb2, len2), len2, dte, dqe, &past);
		if(!past) return ok;
	}
	return wtz_zmer_advance_slow(pb1, len1, pb2, len2, dte, dqe);
}

#endif
