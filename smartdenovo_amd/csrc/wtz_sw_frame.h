/*
 * K-sw3 (kswx_extend_align_shift_core, /root/reference/kswx.h:101-232) on one wavefront, round-5 form: the DP runs in the
 * ANTI-DIAGONAL FRAME  G(i,j) = H(i,j) - (i+j)*E  (same for the E and F states).
 *
 * Why: wtz_extend_shift_reg (wtz_sw_wave.h) spends ~35 VALU ops per cell, a third of them on things that are not the recurrence:
 * the gap-extension adds (e+E, f+E), a per-column multiple of E inside the lane's F aggregate, three per-cell selects that
 * put -10000 into the cells right of the band end, register moves that re-frame the previous row when the band start moves,
 * and a trace-byte computation the compiler sinks into the conditional store blocks (which keeps m, e, f, t of all C cells alive:
 * 256 VGPRs + AGPR copies, one wave per SIMD).  In the frame
 *     m~ = G(i-1,j-1) + s - 2E            E~' = max(E~, m~ + I)            F~' = max(F~, m~ + D)          H~ = max(m~, E~, F~)
 * gap extension costs nothing (a vertical / horizontal step changes i+j by one, which is exactly the E the reference adds), the four
 * decisions of the trace byte are frame-invariant (every comparison is between values of ONE cell), the lane's F aggregate is a plain
 * maximum and the carry-in a plain prefix maximum.  I == D in every caller (wtzmo / wtgbo / wtext pass O for both), so t = m~ + O
 * serves both gap states.  The band start's move (S = 0, 1, 2) is folded into WHICH register a cell reads (three row bodies,
 * in-place register updates in an order that never overwrites an unread input): no moves.  The cells right of the band end are left
 * as they fall (they only feed cells further right); what the reference guarantees about them is restored where it can be observed:
 *   - the three register slots a LATER row can read (H at column je, E at je and je+1: the band end moves by at most 2 per row) are
 *     set to the reference's -10000 after each row - a wave-uniform switch over the register index, three selects per ROW;
 *   - the row maximum: lanes entirely beyond the band end are masked after their local reduction; the one lane the band end cuts
 *     through checks whether its local arg-max fell on a cell beyond it and only then (rare: that cell is W columns off the running
 *     maximum) the wave repeats the reduction with per-cell masks.
 * The trace layout is that of wtz_extend_shift_reg and wtz_shift_traceback is shared; the byte holds the four decisions only - the "bases equal" bit (two ops per
 * cell to insert) is gone: the walk counts diagonal steps and gap runs, and matches / mismatches follow from the score (see wtz_shift_traceback<C, NL, false>).
 * Values are kept < 2^20 in magnitude (caller's envelope, which now includes (ql+tl)*|E| for the frame term) so that the packed
 * arg-max key  H*2048 + (2047 - band column)  fits 32 bits in every intermediate form.
 */
#ifndef WTZ_SW_FRAME_H
#define WTZ_SW_FRAME_H

#include "wtz_sw_wave.h"

#ifdef __HIPCC__

#ifndef WTZ_OCC_EXTFR
#define WTZ_OCC_EXTFR 2
#endif
/* resident waves per SIMD the band classes of the split launch are compiled for (WTZ_EXT_FR_SPLIT=1: bands of <= 16 / <= 28 / <= 32 columns per lane in three
 * concurrent launches, each kernel with the register budget of its widest row body) */
#ifndef WTZ_OCC_EXTFR_LO
#define WTZ_OCC_EXTFR_LO 4
#endif
#ifndef WTZ_OCC_EXTFR_MID
#define WTZ_OCC_EXTFR_MID 2
#endif
#define WTZ_FR_OCC(CHI) ((CHI) <= 16 ? WTZ_OCC_EXTFR_LO : ((CHI) <= 28 ? WTZ_OCC_EXTFR_MID : WTZ_OCC_EXTFR))

/* pins a value in program order: the trace bytes must be computed where the cell is, not sunk into the conditional store block */
#define WTZ_PIN(v) asm volatile("" : "+v"(v))
/* the same with a compile-time tag: bodies of a wave-uniform switch that differ only in their register index stay apart (merged, the index becomes a
 * run-time value and the register array moves to scratch memory) */
#define WTZ_PIN_TAG(v, tag) asm volatile("; slot %1" : "+v"(v) : "n"(tag))

/* f(k) for the wave-uniform k in [LO, HI): a binary tree of scalar branches around straight-line bodies with a compile-time register index */
template<int LO, int HI, typename F>
WTZ_D void wtz_uniform_switch(const int32_t k, F &&f){
	if constexpr(LO + 1 >= HI) f(wtz_ic<LO>{});
	else { constexpr int MID = (LO + HI) / 2; if(k < MID) wtz_uniform_switch<LO, MID>(k, f); else wtz_uniform_switch<MID, HI>(k, f); }
}

template<int C, int S>
WTZ_D void wtz_fr_row(int32_t (&hv)[C], int32_t (&ev)[C], uint32_t (&zw)[(C + 3) / 4], const uint32_t eq_lo, const uint32_t eq_hi, const int lane,
		const int32_t bnd, const int32_t SF, const int32_t MX, const int32_t Xp, const int32_t O, const int32_t (&ck)[C], int32_t &lkey){
	/* ---- pass 1: m~ into hv[] in place (hv[k] <- f(old hv[k + S - 1])), the lane's F aggregate ---- */
	int32_t agg = (int32_t)0x80000000;
	if constexpr(S == 0){
		int32_t prv = wtz_dpp_wave_shr1(0, hv[C - 1]);
		prv = (lane == 0) ? bnd : prv;
		wtz_static_for<0, C>([&](auto kc){
			constexpr int k = C - 1 - decltype(kc)::value;          /* descending: hv[k-1] is still the old value */
			const int32_t b = (int32_t)(((k < 16 ? eq_lo : eq_hi) >> (2 * (k & 15))) & 1u);
			const int32_t src = (k == 0) ? prv : hv[k == 0 ? 0 : k - 1];
			const int32_t m = wtz_mad24(b, MX, src) + Xp;
			hv[k] = m; agg = m > agg ? m : agg;
		});
	} else if constexpr(S == 1){
		wtz_static_for<0, C>([&](auto kc){
			constexpr int k = decltype(kc)::value;
			const int32_t b = (int32_t)(((k < 16 ? eq_lo : eq_hi) >> (2 * (k & 15))) & 1u);
			const int32_t m = wtz_mad24(b, MX, hv[k]) + Xp;
			hv[k] = m; agg = m > agg ? m : agg;
		});
	} else {
		const int32_t nh0 = wtz_dpp_wave_shl1(SF, hv[0]);        /* lane 63: beyond the 64*C columns of the frame, never inside the band */
		wtz_static_for<0, C>([&](auto kc){
			constexpr int k = decltype(kc)::value;                  /* ascending: hv[k+1] is still the old value */
			const int32_t b = (int32_t)(((k < 16 ? eq_lo : eq_hi) >> (2 * (k & 15))) & 1u);
			const int32_t src = (k == C - 1) ? nh0 : hv[k == C - 1 ? k : k + 1];
			const int32_t m = wtz_mad24(b, MX, src) + Xp;
			hv[k] = m; agg = m > agg ? m : agg;
		});
	}
	/* ---- F carry-in: in the frame a horizontal gap costs nothing per column, so the carry is the prefix maximum of (m~ + D) over the lanes to the
	 * left, floored by the frame image SF of the reference's f = -10000 at the band start ---- */
	int32_t f;
	{
		const int32_t pm = wtz_wave_max_scan_excl(agg + O, SF);
		f = pm > SF ? pm : SF;
	}
	/* E~ of the cells: ev[k + S] of the old frame */
	int32_t ne0 = 0, ne1 = 0;
	if constexpr(S >= 1) ne0 = wtz_dpp_wave_shl1(SF, ev[0]);
	if constexpr(S == 2) ne1 = wtz_dpp_wave_shl1(SF, ev[C > 1 ? 1 : 0]);
	int32_t key = (int32_t)0x80000000;
	#pragma unroll
	for(int q4 = 0; q4 < (C + 3) / 4; q4++) zw[q4] = 0;
	wtz_static_for<0, C>([&](auto kc){
		constexpr int k = decltype(kc)::value;
		const int32_t m = hv[k];
		int32_t e;
		if constexpr(S == 0) e = ev[k];
		else if constexpr(S == 1) e = (k == C - 1) ? ne0 : ev[k == C - 1 ? k : k + 1];
		else e = (k == C - 1) ? ne1 : ((k == C - 2) ? ne0 : ev[k >= C - 2 ? k : k + 2]);
		const int32_t h0 = m > e ? m : e;
		uint32_t d = (uint32_t)(m - e) >> 31;                                   /* m < e */
		d = __builtin_amdgcn_alignbit(d, (uint32_t)(h0 - f), 31);               /* max(m,e) < f */
		const int32_t h = h0 > f ? h0 : f;
		const int32_t t = m + O;
		d = __builtin_amdgcn_alignbit(d, (uint32_t)(t - e), 31);                /* E extended: e + E > m + I + E */
		const int32_t en = e > t ? e : t;
		d = __builtin_amdgcn_alignbit(d, (uint32_t)(t - f), 31);                /* F extended */
		f = f > t ? f : t;
		hv[k] = h; ev[k] = en;                                                   /* no "bases equal" bit in this form's trace byte: mat / mis follow from the score (wtz_shift_traceback<.., false>) */
		const int32_t kk = (int32_t)(((uint32_t)h << 11) + (uint32_t)ck[k]);
		key = kk > key ? kk : key;
		zw[k >> 2] |= d << (8 * (k & 3));
		if constexpr((k & 3) == 3 || k == C - 1){ WTZ_PIN(zw[k >> 2]);
#ifdef WTZ_FR_SCHED_BARRIER
			__builtin_amdgcn_sched_barrier(0);
#endif
		}
	});
	lkey = key;
}

template<int C>
WTZ_D wtz_aln_t wtz_extend_shift_fr(int32_t qlen, const wtz_seq_packed &query, int32_t tlen, const wtz_seq_packed &target, int32_t init_score,
		int32_t ql, int32_t tl, int32_t W, int32_t M, int32_t X, int32_t O, int32_t E, int32_t T,
		uint64_t *tb, wtz_trace_t &tr, wtz_pool_t *pool, wtz_cigar_t &cigars, unsigned long long *cells, bool *ok, bool *consistent){
	const int lane = (int)(threadIdx.x & 63);
	wtz_aln_t x; memset(&x, 0, sizeof x);
	*ok = true; *consistent = true;
	if(lane == 0) cigars.n = 0;
	if(init_score < 0) init_score = 0;
	/* the job's numbers arrive through vector loads: say that they are the same in all lanes (round 6: the row loop ran on a VECTOR compare of i against ql - an
	 * exec-masked loop with a dozen mask instructions per row - and kept the band bounds in vector registers) */
	qlen = __builtin_amdgcn_readfirstlane(qlen); tlen = __builtin_amdgcn_readfirstlane(tlen); init_score = __builtin_amdgcn_readfirstlane(init_score);
	ql = __builtin_amdgcn_readfirstlane(ql); tl = __builtin_amdgcn_readfirstlane(tl); W = __builtin_amdgcn_readfirstlane(W);
	M = __builtin_amdgcn_readfirstlane(M); X = __builtin_amdgcn_readfirstlane(X); O = __builtin_amdgcn_readfirstlane(O); E = __builtin_amdgcn_readfirstlane(E); T = __builtin_amdgcn_readfirstlane(T);
	constexpr int C4 = (C + 3) / 4;
	const uint32_t zrow = (uint32_t)C4 * 256u;
	if(!wtz_trace_prepare(tr, pool, zrow, ql, true)){ *ok = false; return x; }
	uint8_t **zchunk = tr.chunk; int32_t *zb = tr.zb;
	uint8_t *z = NULL;
	{
		const int32_t nw = (tl + 31) / 32 + 1;
		for(int32_t w = lane; w < nw; w += 64) tb[w] = wtz_pack32(target, w * 32, tl);
	}
	__threadfence_block();
	const int32_t colrel0 = lane * C;
	const int32_t MX = M - X, Xp = X - 2 * E;
	/* arg-max key of cell k before the lane / row term: G*2048 + k*(2048*E - 1); wave-uniform, meant for scalar registers */
	int32_t ck[C];
	#pragma unroll
	for(int k = 0; k < C; k++) ck[k] = __builtin_amdgcn_readfirstlane(k * (2048 * E - 1));
	/* row 0 as a row with S = 1 over a synthetic row -1 whose frame starts at column -1:
	 * slot p of lane l = column c = l*C + p - 1;  H(-1,c) = rh[c+1] (kswx.h:143-144),  E(0,c) = -10000 (kswx.h:145) */
	int32_t hv[C], ev[C];
	#pragma unroll
	for(int p = 0; p < C; p++){
		const int32_t c = colrel0 + p - 1;
		const int32_t hr = (c < 0) ? init_score : init_score + O + E * (c + 1);
		hv[p] = hr - (c - 1) * E;             /* G(-1,c) = H - (i+j)E with i = -1 */
		ev[p] = -10000 - c * E;               /* E~(0,c) */
	}
	int32_t mx = init_score, mi = -1, mj = -1, gmax = 0, gi = -1, gj = -1;
	int32_t jbp = -1, c = 0, i;
	unsigned long long ncell = 0;
	uint32_t qw_lo = 0, qw_hi = 0, qcur = 0;
	int32_t jb_n = 0, je_n = tl; uint64_t tbits_n;
	/* arg-max key offset of the lane: ((i + jb + l*C) * E) * 2048 + 2047 - l*C; it moves by (1 + band shift) * E * 2048 per row (round 6: was a multiply per row) */
	int32_t koff = (colrel0 * E) * 2048 + 2047 - colrel0;
	const int32_t E2048 = E * 2048;
	/* band start of row r in lane r & 63 (v_writelane), written out once per 64 rows (round 6: lane 0 stored one word per row behind an exec-mask switch) */
	int32_t zbv = 0;
	{
		if(je_n > W + 1) je_n = W + 1;              /* row 0: c = 0 */
		if(je_n > tl) je_n = tl;
		const int32_t jj = colrel0 < tl ? colrel0 : (tl > 0 ? tl - 1 : 0);
		const int32_t w = jj >> 5, sh = (jj & 31) * 2;
		const uint64_t w0 = tb[w], w1 = tb[w + 1];
		tbits_n = sh ? ((w0 >> sh) | (w1 << (64 - sh))) : w0;
	}
	__builtin_amdgcn_s_waitcnt(0x0F70);
	for(i = 0; i < ql; i++){
		if((i & 63) == 0){
			/* every branch of this block is wave-uniform by construction and the block ends in an explicit vmcnt(0): see wtz_extend_shift_reg */
			const uint32_t ci = (uint32_t)i >> 6;
			unsigned long long za = 0;
			const int have = __builtin_amdgcn_readfirstlane(ci < tr.n_chunk ? 1 : 0);
			if(have) za = (unsigned long long)(uintptr_t)wtz_as_global(zchunk)[ci];
			else {
				if(lane == 0){ uint8_t *p = (uint8_t*)wtz_pool_alloc(pool, (size_t)zrow * 64); wtz_as_global(zchunk)[ci] = p; za = (unsigned long long)(uintptr_t)p; }
				za = __shfl(za, 0, 64);
			}
			const uint32_t zlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)za), zhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(za >> 32));
			z = (uint8_t*)(uintptr_t)(((unsigned long long)zhi << 32) | zlo);
			if((zlo | zhi) == 0){ *ok = false; break; }
			if(!have) tr.n_chunk = ci + 1;
			if((i & 2047) == 0){ const uint64_t qw = wtz_pack32(query, i + lane * 32, ql); qw_lo = (uint32_t)qw; qw_hi = (uint32_t)(qw >> 32); }
			__builtin_amdgcn_s_waitcnt(0x0F70);
		}
		const int32_t jb = jb_n, je = je_n;
		if((i & 15) == 0){
			const int32_t qs = __builtin_amdgcn_readfirstlane((i & 2047) >> 5);
			qcur = (i & 16) ? (uint32_t)__builtin_amdgcn_readlane((int)qw_hi, qs) : (uint32_t)__builtin_amdgcn_readlane((int)qw_lo, qs);
		}
		const uint32_t qbase = (qcur >> ((i & 15) * 2)) & 3u;
		const uint64_t tbits = tbits_n;
		uint32_t eq_lo, eq_hi;
		{
			const uint32_t qrep = 0x55555555u * qbase;
			const uint32_t x_lo = (uint32_t)tbits ^ qrep, x_hi = (uint32_t)(tbits >> 32) ^ qrep;
			eq_lo = ~(x_lo | (x_lo >> 1)) & 0x55555555u; eq_hi = ~(x_hi | (x_hi >> 1)) & 0x55555555u;
		}
		const int32_t s = __builtin_amdgcn_readfirstlane(jb - jbp);
		/* frame images of the reference's boundary values of this row:
		 *   bnd = G(i-1, jb-1): H(i-1,-1) = init + I + E*i when jb == 0 (kswx.h:152), -10000 otherwise (outside the previous band)
		 *   SF  = F~ at the band start: f = -10000 (kswx.h:156) */
		const int32_t bnd = ((jb == 0) ? init_score + O + E * i : -10000) - (i + jb - 2) * E;
		const int32_t SF = -10000 - (i + jb) * E;
		uint32_t zw[C4]; int32_t key;
		if(s == 1)      wtz_fr_row<C, 1>(hv, ev, zw, eq_lo, eq_hi, lane, bnd, SF, MX, Xp, O, ck, key);
		else if(s == 0) wtz_fr_row<C, 0>(hv, ev, zw, eq_lo, eq_hi, lane, bnd, SF, MX, Xp, O, ck, key);
		else            wtz_fr_row<C, 2>(hv, ev, zw, eq_lo, eq_hi, lane, bnd, SF, MX, Xp, O, ck, key);
		const int32_t nvt = je - jb;                       /* band-relative column of the first cell beyond the band end */
		/* ---- row maximum and its FIRST arg-max (kswx.h:172) ---- */
		{
			const bool part = colrel0 < nvt && colrel0 + C > nvt;
			const int32_t best_col = 2047 - ((key + koff) & 2047);      /* band-relative column of the lane's arg-max */
			const bool redo = part && best_col >= nvt;
			if(__builtin_amdgcn_readfirstlane((int)(__ballot(redo) != 0ull))){
				int32_t k2 = (int32_t)0x80000000;
				#pragma unroll
				for(int k = 0; k < C; k++){
					const int32_t kk = (int32_t)(((uint32_t)hv[k] << 11) + (uint32_t)ck[k]);
					const int32_t km = (colrel0 + k < nvt) ? kk : (int32_t)0x80000000;
					k2 = km > k2 ? km : k2;
				}
				key = k2;
			}
			key = (colrel0 < nvt) ? key + koff : (int32_t)0x80000000;
		}
		ncell += (unsigned long long)(je - jb);
		key = wtz_wave_max_i32(key);
		int32_t imax = 0, mj2 = -1;
		if((key >> 11) > 0){ imax = key >> 11; mj2 = jb + (2047 - (key & 2047)); }
		zbv = (lane == (i & 63)) ? jb : zbv;      /* (v_writelane would take two scalar operands: one more than a VALU instruction of this ISA may read) */
		if((i & 63) == 63) wtz_as_global(zb)[(i & ~63) + lane] = zbv;
		/* ---- H(i, je-1) for the target-end rule ---- */
		if(je == tlen){
			const int32_t idx = nvt - 1;
			const int32_t Ll = __builtin_amdgcn_readfirstlane(idx / C), kl = __builtin_amdgcn_readfirstlane(idx % C);
			int32_t hsel = 0;
			wtz_uniform_switch<0, C>(kl, [&](auto kc){ constexpr int k = decltype(kc)::value; hsel = hv[k]; WTZ_PIN_TAG(hsel, k); });
			const int32_t h1 = __builtin_amdgcn_readlane(hsel, Ll) + (i + je - 1) * E;      /* H(i, je-1) */
			if(gmax < h1){ gmax = h1; gi = i; gj = je - 1; }
		}
		if(i + 1 == qlen && gmax < imax){ gmax = imax; gi = i; gj = mj2; }
		jbp = jb;
		bool stop = false;
		if(imax > mx){ mx = imax; mi = i; mj = mj2; }
		else if(imax <= 0) stop = true;
		if(!stop){
			c++; if(c < mj2) c++; else if(c > mj2) c--;
			jb_n = 0; je_n = tl;
			if(jb_n < c - W) jb_n = c - W;
			if(je_n > c + W + 1) je_n = c + W + 1;
			if(je_n > tl) je_n = tl;
			koff += (1 + jb_n - jb) * E2048;
			if(jb_n != jb){       /* the band start moved: the lane's target bits move with it (round 6: while the band start stands - the first W rows of every job - they are kept) */
				const int32_t j0n = jb_n + colrel0;
				const int32_t jj = j0n < tl ? j0n : (tl > 0 ? tl - 1 : 0);
				const int32_t w = jj >> 5, sh = (jj & 31) * 2;
				const uint64_t w0 = tb[w], w1 = tb[w + 1];
				tbits_n = sh ? ((w0 >> sh) | (w1 << (64 - sh))) : w0;
			}
			/* ---- the slots the NEXT row reads beyond this row's band end: H(i, je), E(i+1, je), E(i+1, je+1) are the reference's -10000 (kswx.h:179, 191-192).  They are
			 * read only by a row whose band end lies further right (by one or two columns), so they are set only then (round 6: every row paid four scalar divisions,
			 * a switch over the register index and four selects; a band that stands still - every row of a job whose target side is shorter than W, every row after
			 * the band has reached the target's end - needs none of it: what lies beyond its end is never read by a cell inside it) ---- */
			if(je_n > je){
				const int32_t Lb = __builtin_amdgcn_readfirstlane(nvt / C), kb = __builtin_amdgcn_readfirstlane(nvt % C);
				const int32_t SG = -10000 - (i + je) * E;               /* rh[je+1] = -10000 (kswx.h:191): H(i, je) */
				const int32_t SE1 = -10000 - (i + 1 + je) * E;          /* re[je] = -10000 (kswx.h:179): E(i+1, je) */
				const int32_t SE2 = -10000 - (i + 2 + je) * E;          /* re[je+1] = -10000 (kswx.h:192): E(i+1, je+1) */
				if(Lb < 64){
					wtz_uniform_switch<0, C>(kb, [&](auto kc){
						constexpr int k = decltype(kc)::value;
						hv[k] = (lane == Lb) ? SG : hv[k];
						ev[k] = (lane == Lb) ? SE1 : ev[k];
						if constexpr(k + 1 < C) ev[k + 1] = (lane == Lb) ? SE2 : ev[k + 1];
						else ev[0] = (lane == Lb + 1) ? SE2 : ev[0];
						WTZ_PIN_TAG(hv[k], k);
					});
				}
			}
		}
#ifndef WTZ_EXP_NOTRACE
		if(colrel0 < nvt){
			WTZ_GLOBAL_AS uint32_t *zr = wtz_as_global((uint32_t*)(z + (size_t)(i & 63) * zrow) + lane);
			#pragma unroll
			for(int q4 = 0; q4 < C4; q4++) zr[(size_t)q4 * 64] = zw[q4];
		}
#endif
#ifdef WTZ_EXP_ROWS1
		stop = true;       /* diagnostic build: one row per job (the fixed cost of a job; results are wrong) */
#endif
		if(stop) break;
	}
	if(cells && lane == 0) *cells += ncell;
	if(!*ok) return x;
	{   /* band starts of the rows of the last, incomplete block of 64 (i = rows run; a `break` leaves i at the last row run) */
		const int32_t last = i < ql ? i : ql - 1;
		if(last >= 0 && (last & 63) != 63 && lane <= (last & 63)) wtz_as_global(zb)[(last & ~63) + lane] = zbv;
	}
	if(gmax > 0 && gmax >= mx + T){ x.score = gmax; x.qe = gi; x.te = gj; }
	else { x.score = mx; x.qe = mi; x.te = mj; }
	__threadfence_block();
	const wtz_tb_score sc = { M, X, O, E, init_score };
	if(!wtz_shift_traceback<C, 64, false>(x, zchunk, zb, zrow, tb, cigars, &sc)) *ok = false, *consistent = false;
	return wtz_bcast_aln(x);
}

/* one K-sw3 job on the calling wavefront in the frame form; false = the job is outside the envelope (or its counts did not follow from the score) and stays
 * open for the older forms (wtz_kernel_extjobs_reg / wtz_kernel_extjobs).  stb: TW 64-bit words of LDS of this wave. */
template<int TW, int CLO, int CHI>
WTZ_D bool wtz_extjob_run_fr(wtz_extjob_t *job, const wtz_params_t *Pm, wtz_pool_t *pool, wtz_pool_t *tpool, uint64_t *stb){
	if(!job->valid || job->done) return true;
	const int lane = (int)(threadIdx.x & 63);
	if(job->qlen <= 0 || job->tlen <= 0) return false;
	const int32_t init_score = job->init_score < 0 ? 0 : job->init_score;
	int32_t W = job->W, ql, tl, n_col;
	wtz_ext_geometry(job->qlen, job->tlen, init_score, W, Pm->M, Pm->O, Pm->O, Pm->E, Pm->T, ql, tl, n_col);
	const int32_t Cw = (n_col + 63) / 64;
	if(Cw > 32 || (tl + 63) / 32 + 1 > TW || (ql + 63) / 64 > WTZ_TRACE_MAXCHUNK) return false;
	{
		/* every value of the DP, its frame image and the -10000 family stay below 2^20 in magnitude */
		const long long aE = Pm->E < 0 ? -(long long)Pm->E : (long long)Pm->E, aX = Pm->X < 0 ? -(long long)Pm->X : (long long)Pm->X, aO = Pm->O < 0 ? -(long long)Pm->O : (long long)Pm->O;
		const long long span = (long long)ql + tl + 4;
		if((long long)init_score + (long long)(Pm->M > 0 ? Pm->M : -Pm->M) * (ql < tl ? ql : tl) + span * aE + 10000 + aX + aO + 16 >= (1 << 20)) return false;
		if(aE > 255 || Pm->M == Pm->X) return false;
	}
	if(Cw <= CLO || (CHI < 32 && Cw > CHI)) return false;
	wtz_trace_t tr; tr.chunk = NULL; tr.zb = NULL; tr.n_chunk = 0; tr.zrow = 0; tr.cap_rows = 0;
	wtz_cigar_t cg; cg.a = NULL; cg.n = cg.cap = 0; cg.pool = pool; cg.bad = 0;
	if(lane == 0) cg.init(pool, (uint32_t)ql / 2u + 16u);
	unsigned long long cells = 0; bool ok = true, consistent = true;
	wtz_aln_t x;
#define WTZ_EXTFR_CASE(CM) x = wtz_extend_shift_fr<CM>(job->qlen, job->q, job->tlen, job->t, job->init_score, ql, tl, W, Pm->M, Pm->X, Pm->O, Pm->E, Pm->T, stb, tr, tpool, cg, &cells, &ok, &consistent)
	/* one instantiation per two columns per lane from 4 on (round 6: steps of four left the mean job a sixth of its cells beyond its band's widest row) */
	if(Cw <= 4){ if(CLO < 4 && CHI >= 4) WTZ_EXTFR_CASE(4); }
	else if(Cw <= 6){ if(CLO < 6 && CHI >= 6) WTZ_EXTFR_CASE(6); }
	else if(Cw <= 8){ if(CLO < 8 && CHI >= 8) WTZ_EXTFR_CASE(8); }
	else if(Cw <= 10){ if(CLO < 10 && CHI >= 10) WTZ_EXTFR_CASE(10); }
	else if(Cw <= 12){ if(CLO < 12 && CHI >= 12) WTZ_EXTFR_CASE(12); }
	else if(Cw <= 14){ if(CLO < 14 && CHI >= 14) WTZ_EXTFR_CASE(14); }
	else if(Cw <= 16){ if(CLO < 16 && CHI >= 16) WTZ_EXTFR_CASE(16); }
	else if(Cw <= 18){ if(CLO < 18 && CHI >= 18) WTZ_EXTFR_CASE(18); }
	else if(Cw <= 20){ if(CLO < 20 && CHI >= 20) WTZ_EXTFR_CASE(20); }
	else if(Cw <= 22){ if(CLO < 22 && CHI >= 22) WTZ_EXTFR_CASE(22); }
	else if(Cw <= 24){ if(CLO < 24 && CHI >= 24) WTZ_EXTFR_CASE(24); }
	else if(Cw <= 26){ if(CLO < 26 && CHI >= 26) WTZ_EXTFR_CASE(26); }
	else if(Cw <= 28){ if(CLO < 28 && CHI >= 28) WTZ_EXTFR_CASE(28); }
	else { if(CHI >= 32) WTZ_EXTFR_CASE(32); }
#undef WTZ_EXTFR_CASE
	if(!consistent) return false;          /* mat / mis did not follow from the score: the job stays open for the general kernel (wtz_kernel_extjobs) */
	if(lane == 0){ job->x = x; job->cigar = cg.a; job->cigar_len = cg.n; job->bad = (!ok || cg.bad); job->cells = cells; job->done = 5; }
	return true;
}

/* one wavefront per job */
template<int TW, int CLO = 0, int CHI = 32>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WTZ_FR_OCC(CHI), 8))) wtz_kernel_extjobs_fr(wtz_extjob_t *jobs, const uint32_t *order, uint32_t n, const wtz_params_t *Pm, wtz_pool_t *pool, wtz_pool_t *tpool){
	__shared__ uint64_t stb[TW];
	const uint32_t b = blockIdx.x;
	if(b >= n) return;
	(void)wtz_extjob_run_fr<TW, CLO, CHI>(&jobs[order ? order[b] : b], Pm, pool, tpool, stb);
}

#endif /* __HIPCC__ */
#endif
