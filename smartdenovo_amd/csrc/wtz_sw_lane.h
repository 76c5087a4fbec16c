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
 * skipped.  Trace: 4 bits per cell (bits 1:0 source of H, bit 2 E extended, bit 3 F extended, as in wtz_sw.h), rows contiguous PER LANE
 * in the transient pool — a row is one to four 16-byte stores per lane and the traceback of a lane walks its own cache lines.
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
 * RS dwords), runs: room for ql + tl + 2 runs, written in TRACEBACK order (the reader reverses).
 */
template<int NC, bool ABS>
WTZ_HD void wtz_lane_fixed(bool live, int32_t qlen, const wtz_seq_packed &q, int32_t tlen, const wtz_seq_packed &t, int32_t init_score,
		int32_t W, int32_t ql, int32_t tl, int32_t M, int32_t X, int32_t I, int32_t D, int32_t E, int32_t T, uint32_t *tr, uint32_t *runs, wtz_lres_t &R){
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
		int32_t h1 = jb == 0 ? a + I + E * (i + 1) : WTZ_LN_NEG, f = WTZ_LN_NEG, imax = ABS ? 0 : WTZ_LN_LOW, mj2 = -1;     /* kswx.h:274-280 */
		const int32_t nmax = wtz_lane_wmax(n);
		uint32_t acc[4] = {0u, 0u, 0u, 0u};
		uint32_t *trow = tr + (size_t)i * RS;
		#pragma unroll
		for(int c0 = 0; c0 < NC; c0 += 8){
			if(c0 <= nmax){                                   /* uniform: some lane of the wave has a cell (or its closing slot) in this block of 8 columns */
				#pragma unroll
				for(int c = c0; c < c0 + 8 && c < NC; c++){
					if(c < n){
						const int32_t hd = S ? RH[c + 1] : RH[c], ev = S ? RE[c + 1] : RE[c];
						const uint32_t tb = (TW[c >> 4] >> (2 * (c & 15))) & 3u;
						const int32_t m = hd + (tb == qb ? M : X);
						RH[c] = h1;
						uint32_t d = m >= ev ? 0u : 1u;
						int32_t h = m >= ev ? m : ev;
						d = h >= f ? d : 2u;
						h = h >= f ? h : f;
						h1 = h;
						mj2 = imax > h ? mj2 : jb + c;                 /* LAST arg-max (kswx.h:288-289) */
						imax = imax > h ? imax : h;
						int32_t tt = m + I + E; const int32_t e2 = ev + E;
						d |= e2 > tt ? 4u : 0u;
						RE[c] = e2 > tt ? e2 : tt;
						tt = m + D + E; const int32_t f2 = f + E;
						d |= f2 > tt ? 8u : 0u;
						f = f2 > tt ? f2 : tt;
						acc[(c >> 3) & 3] |= d << (4 * (c & 7));
					} else if(c == n){ RH[c] = h1; RE[c] = WTZ_LN_NEG; }        /* rh[je] = h1; re[je] = -10000 (kswx.h:298) */
				}
			}
			if(((c0 + 8) & 31) == 0 || c0 + 8 >= NC){          /* a group of 32 columns is complete: one 16-byte store */
				if(on && (c0 & ~31) < n){
					if(RS == 2){ trow[0] = acc[0]; trow[1] = acc[1]; }
					else { uint32_t *p = trow + (c0 >> 5) * 4; p[0] = acc[0]; p[1] = acc[1]; p[2] = acc[2]; p[3] = acc[3]; }
				}
				acc[0] = acc[1] = acc[2] = acc[3] = 0u;
			}
		}
		if(n == NC){ RH[NC] = h1; RE[NC] = WTZ_LN_NEG; }
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
	int32_t i_ = useg ? gi : bi, j_ = useg ? gj : bj;
	R.score = useg ? gmax : best; R.qe = i_ + 1; R.te = j_ + 1;
	R.minrow = minrow; R.flags = (useg ? WTZ_LR_USEDG : 0u) | WTZ_LR_DONE; R.cells = ncell;
	/* traceback (kswx.h:309-333): each lane walks its own rows */
	int32_t mat = 0, mis = 0, ins = 0, del = 0, qblk = -1, tblk = -1; uint64_t tw = 0; qw = 0;
	uint32_t state = 0;
	wtz_lruns_t Wr; Wr.a = runs; Wr.n = 0; Wr.tail = 0;
	if(!live){ i_ = -1; j_ = -1; }
	while(i_ >= 0 && j_ >= 0){
		const int32_t jb = i_ > W ? i_ - W : 0, c = j_ - jb;
		const uint32_t nib = (tr[(size_t)i_ * RS + (c >> 3)] >> (4 * (c & 7))) & 15u;
		state = state == 0 ? (nib & 3u) : (state == 1 ? ((nib & 4u) ? 1u : 0u) : ((nib & 8u) ? 2u : 0u));
		if((i_ >> 5) != qblk){ qblk = i_ >> 5; qw = wtz_pack32(q, qblk * 32, qlen); }
		if((j_ >> 5) != tblk){ tblk = j_ >> 5; tw = wtz_pack32(t, tblk * 32, tlen); }
		if(state == 0){
			if(((qw >> (2 * (i_ & 31))) & 3ull) == ((tw >> (2 * (j_ & 31))) & 3ull)) mat++; else mis++;
			i_--; j_--;
		} else if(state == 1){ i_--; ins++; }
		else { j_--; del++; }
		wtz_lruns_push(Wr, state, 1);
	}
	if(live && i_ >= 0){ ins += i_ + 1; wtz_lruns_push(Wr, 1, (uint32_t)(i_ + 1)); }
	if(live && j_ >= 0){ del += j_ + 1; wtz_lruns_push(Wr, 2, (uint32_t)(j_ + 1)); }
	wtz_lruns_finish(Wr);
	R.mat = mat; R.mis = mis; R.ins = ins; R.del = del; R.n_runs = Wr.n;
}

/* hzm_aln.h:278-314 without the CIGAR: where the z-mer run alignment leaves the cursor (te, qe), and whether the pair aligns at all */
template<typename S1, typename S2>
WTZ_HD bool wtz_zmer_advance(const S1 &pb1, uint32_t len1, const S2 &pb2, uint32_t len2, int32_t *dte, int32_t *dqe){
	uint32_t s0 = 0, s1 = 0;
	while(s0 < len1 || s1 < len2){
		const uint32_t b = pb1.at((int32_t)s0);
		if(b != pb2.at((int32_t)s1)) return false;
		uint32_t e0 = s0 + 1; while(e0 < len1 && pb1.at((int32_t)e0) == b) e0++;
		uint32_t e1 = s1 + 1; while(e1 < len2 && pb2.at((int32_t)e1) == b) e1++;
		s0 = e0; s1 = e1;
	}
	*dte = (int32_t)s0; *dqe = (int32_t)s1;      /* te = mat + del = sum of l0, qe = mat + ins = sum of l1 */
	return true;
}

#endif
