/*
 * K-sw3 (kswx_extend_align_shift_core, /root/reference/kswx.h:101-232) on one wavefront in the anti-diagonal frame of wtz_sw_frame.h, with TWO
 * cells per vector register (round 6): the frame values as 16-bit halves, the row body in packed instructions (v_pk_add / v_pk_max / v_pk_sub ... clamp).
 *
 * Why: wtz_extend_shift_fr runs at two thirds of the VALU issue rate of the chip (PMC: 162 G wave-instructions per configs[2] step on 253 G issue slots), 20
 * instructions per computed cell - the kernel is bound by its instruction count, not by occupancy or memory.  A packed instruction computes two cells.
 *
 * Values.  Every in-band value of the frame lies in [lo, hi] with
 *     hi = init + M*min(ql, tl) + (ql + tl)*|E| (+ margins),     lo = -10000 + (ql + 2)*min(0, X - 2E) + min(0, O) (+ margins)
 * (H never falls by more than |X - 2E| per row below the lowest value of the row above, the -10000 family only rises in the frame; E~, F~ and t lie at most
 * |O| under an H).  A job whose window hi - lo fits 16 bits keeps  value - (hi + lo)/2  in signed halves: sums never wrap inside the band, comparisons are
 * signed 16-bit maxima, and the four decisions are the SIGNS of saturating differences (v_pk_sub_i16 clamp: the sign survives a difference beyond 15 bits).
 * Cells right of the band end may wrap - they feed only cells further right (wtz_sw_frame.h) and the row maximum excludes them EXACTLY here (below), not by
 * "a cell W columns off the maximum never wins".  Where the init score alone pushes the window beyond 16 bits the -10000 family is raised (wtz_pk_window,
 * case (b): the argument is stated there).  A job outside both windows is declined and stays open for wtz_extend_shift_fr.
 *
 * Layout.  Lane l owns the C = 2*C2 band-relative columns l*C .. l*C + C - 1 as two runs: register k holds column l*C + k in its low half (run A) and column
 * l*C + C2 + k in its high half (run B).  Both runs move through the three row bodies (band shift S = 0, 1, 2) by register renaming exactly like the 32-bit
 * form; the run edges take one v_alignbit each: run A's left neighbour is run B of the lane before, run B's left neighbour is run A of the own lane.
 * The F chain runs through both runs at once; run B's carry-in is run A's carry-in joined with run A's aggregate (pass 1 delivers both aggregates in one
 * packed maximum), the lanes' carry-in is the prefix maximum over lane aggregates as before.
 * Row maximum: packed maxima of  h + column*E  over groups of four registers, the lane's two run maxima into one 32-bit key (value, 127 - run index), one wave
 * reduction; the run the band end cuts through contributes the maximum over its valid cells only (a wave-uniform switch over the cut position); the FIRST
 * arg-max column inside the winning run comes from per-lane match masks (a 16-bit compare on one half + an add with carry per register: first the group
 * maxima, then the four registers of one group), of which only the winning lane's two words go to the scalar unit.
 * Target: two bit planes in LDS (low / high bit of the base, 32 columns per word), so that "bases equal" is one dense bit per column: (P_lo ^ ~q_lo) & (P_hi ^ ~q_hi).
 * Trace: a NIBBLE per cell (the four decisions), eight cells per dword: half the trace bytes of the 32-bit form; wtz_shift_traceback_pk stages it into the
 * walker's byte window of wtz_shift_traceback.
 */
#ifndef WTZ_SW_FRAME16_H
#define WTZ_SW_FRAME16_H

#include "wtz_sw_frame.h"

#ifdef __HIPCC__

#ifndef WTZ_OCC_EXTPK
#define WTZ_OCC_EXTPK 4
#endif

typedef short wtz_v2s __attribute__((ext_vector_type(2)));
WTZ_D uint32_t wtz_pk_max(uint32_t a, uint32_t b){ return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(wtz_v2s, a), __builtin_bit_cast(wtz_v2s, b))); }
/* a + b per half, saturating: inside a job's window it is the plain sum; the family that stands in for minus infinity stops at the bottom of the 16 bits instead of wrapping */
WTZ_D uint32_t wtz_pk_adds(uint32_t a, uint32_t b){ return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(wtz_v2s, a), __builtin_bit_cast(wtz_v2s, b))); }
/* sign of a - b in each half, whatever the distance (saturating difference) */
WTZ_D uint32_t wtz_pk_subs(uint32_t a, uint32_t b){ return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(wtz_v2s, a), __builtin_bit_cast(wtz_v2s, b))); }
/* a*b + c per half (mod 2^16) in one op */
WTZ_D uint32_t wtz_pk_mad(uint32_t a, uint32_t b, uint32_t c){ uint32_t r; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
/* (a & mask) | (b & ~mask): v_bfi_b32 */
WTZ_D uint32_t wtz_bfi(uint32_t mask, uint32_t a, uint32_t b){ uint32_t r; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(mask), "v"(a), "v"(b)); return r; }      /* (the compiler's own pattern comes out as v_and + v_and_or) */
WTZ_D uint32_t wtz_pk2(int32_t lo, int32_t hi){ return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); }
WTZ_D int32_t wtz_pk_lo(uint32_t a){ return (int32_t)(int16_t)(a & 0xFFFFu); }
WTZ_D int32_t wtz_pk_hi(uint32_t a){ return (int32_t)a >> 16; }
/* (lo >> 16) | (hi << 16): the high half of `lo` under the low half of `hi` */
WTZ_D uint32_t wtz_pk_join(uint32_t hi, uint32_t lo){ return __builtin_amdgcn_alignbit(hi, lo, 16); }
/* acc*2 + (half H of v == low half of t): a compare on one half (SDWA) and an add with carry */
template<int H>
WTZ_D uint32_t wtz_pk_eqacc(uint32_t acc, uint32_t v, uint32_t t){
	if constexpr(H) asm("v_cmp_eq_u16_sdwa vcc, %1, %2 src0_sel:WORD_1 src1_sel:WORD_0\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(v), "v"(t) : "vcc");
	else asm("v_cmp_eq_u16_sdwa vcc, %1, %2 src0_sel:WORD_0 src1_sel:WORD_0\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(v), "v"(t) : "vcc");
	return acc;
}
/* the even bits of a 64-bit word, packed */
WTZ_D uint32_t wtz_even_bits(uint64_t x){
	x &= 0x5555555555555555ULL;
	x = (x | (x >> 1)) & 0x3333333333333333ULL; x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0FULL; x = (x | (x >> 4)) & 0x00FF00FF00FF00FFULL;
	x = (x | (x >> 8)) & 0x0000FFFF0000FFFFULL; x = (x | (x >> 16)) & 0x00000000FFFFFFFFULL;
	return (uint32_t)x;
}

/* one DP row.  hv / ev: H~ and E~ of the previous row in the previous frame on entry, of this row in this frame on exit; zw: the row's trace nibbles;
 * gm: maxima of h + column*E over groups of four registers; eqw: "bases equal" of run A in bits 0..C2-1, of run B in bits 16..16+C2-1;
 * bndp: the frame image of H(i-1, jb-1) in the HIGH half (what lane 0's run A reads through the lane shift); SFp: F~ at the band start in both halves */
template<int C2, int S>
WTZ_D void wtz_pk_row(uint32_t (&hv)[C2], uint32_t (&ev)[C2], uint32_t (&zw)[(C2 + 3) / 4], uint32_t (&gm)[(C2 + 3) / 4], const uint32_t eqw, const uint32_t bndp, const uint32_t SFp,
		const int32_t SF, const int32_t O, const uint32_t MXp, const uint32_t Xpp, const uint32_t Op, const uint32_t (&ckp)[C2]){
	/* ---- pass 1: m~ into hv[] in place, the runs' F aggregates ---- */
	uint32_t agg = 0x80008000u;
	auto cell1 = [&](auto kc, const uint32_t src) -> uint32_t {
		constexpr int k = decltype(kc)::value;
		const uint32_t b = (k ? (eqw >> k) : eqw) & 0x00010001u;
		const uint32_t m = wtz_pk_adds(wtz_pk_mad(b, MXp, src), Xpp);
		agg = wtz_pk_max(agg, m);
		return m;
	};
	if constexpr(S == 0){
		const uint32_t prv = (uint32_t)wtz_dpp_wave_shr1((int32_t)bndp, (int32_t)hv[C2 - 1]);
		const uint32_t src0 = wtz_pk_join(hv[C2 - 1], prv);      /* run A: run B's last cell of the lane before; run B: run A's last cell */
		wtz_static_for<0, C2>([&](auto kc){
			constexpr int k = C2 - 1 - decltype(kc)::value;          /* descending: hv[k-1] is still the old value */
			hv[k] = cell1(wtz_ic<k>{}, (k == 0) ? src0 : hv[k == 0 ? 0 : k - 1]);
		});
	} else if constexpr(S == 1){
		wtz_static_for<0, C2>([&](auto kc){ constexpr int k = decltype(kc)::value; hv[k] = cell1(kc, hv[k]); });
	} else {
		const uint32_t nxt = (uint32_t)wtz_dpp_wave_shl1((int32_t)SFp, (int32_t)hv[0]);      /* lane 63's run B ends beyond the frame, never inside the band */
		const uint32_t srcl = wtz_pk_join(nxt, hv[0]);           /* run A: run B's first cell; run B: run A's first cell of the next lane */
		wtz_static_for<0, C2>([&](auto kc){
			constexpr int k = decltype(kc)::value;                  /* ascending: hv[k+1] is still the old value */
			hv[k] = cell1(kc, (k == C2 - 1) ? srcl : hv[k == C2 - 1 ? k : k + 1]);
		});
	}
	/* ---- F carry-in: prefix maximum of (m~ + D) over the runs to the left, floored by SF ---- */
	uint32_t f;
	{
		const int32_t aA = wtz_pk_lo(agg), aB = wtz_pk_hi(agg);
		const int32_t aL = aA > aB ? aA : aB;
		const int32_t pm = wtz_wave_max_scan_excl(aL + O, SF);
		const int32_t fA = pm > SF ? pm : SF;
		const int32_t fB = fA > aA + O ? fA : aA + O;
		f = wtz_pk2(fA, fB);
	}
	/* E~ of the cells: ev[k + S] of the old frame; the runs' last S cells come from the next run */
	uint32_t ne0 = 0, ne1 = 0;
	if constexpr(S >= 1) ne0 = wtz_pk_join((uint32_t)wtz_dpp_wave_shl1((int32_t)SFp, (int32_t)ev[0]), ev[0]);
	if constexpr(S == 2) ne1 = wtz_pk_join((uint32_t)wtz_dpp_wave_shl1((int32_t)SFp, (int32_t)ev[C2 > 1 ? 1 : 0]), ev[C2 > 1 ? 1 : 0]);
	wtz_static_for<0, C2>([&](auto kc){
		constexpr int k = decltype(kc)::value;
		const uint32_t m = hv[k];
		uint32_t e;
		if constexpr(S == 0) e = ev[k];
		else if constexpr(S == 1) e = (k == C2 - 1) ? ne0 : ev[k == C2 - 1 ? k : k + 1];
		else e = (k == C2 - 1) ? ne1 : ((k == C2 - 2) ? ne0 : ev[k >= C2 - 2 ? k : k + 2]);
		const uint32_t h0 = wtz_pk_max(m, e);
		const uint32_t s0 = wtz_pk_subs(m, e);                                     /* sign: m < e */
		const uint32_t s1 = wtz_pk_subs(h0, f);                                    /* max(m,e) < f */
		const uint32_t h = wtz_pk_max(h0, f);
		const uint32_t t = wtz_pk_adds(m, Op);
		const uint32_t s2 = wtz_pk_subs(t, e);                                     /* E extended */
		const uint32_t en = wtz_pk_max(e, t);
		const uint32_t s3 = wtz_pk_subs(t, f);                                     /* F extended */
		f = wtz_pk_max(f, t);
		hv[k] = h; ev[k] = en;
		/* the four signs into bits 15..12 of each half (two ops per merge: a shift and a v_bfi), then into the register's nibble of the row's trace word; bits
		 * outside a nibble are never kept (the first register of a word brings garbage into the nibbles the next three overwrite; a last, incomplete word keeps it
		 * in nibbles nobody reads) */
		const uint32_t x01 = wtz_bfi(0x80008000u, s0, s1 >> 1), x23 = wtz_bfi(0x80008000u, s2, s3 >> 1);
		const uint32_t x = wtz_bfi(0xC000C000u, x01, x23 >> 2);
		const uint32_t v = wtz_pk_adds(h, ckp[k]);
		if constexpr((k & 3) == 0){ gm[k >> 2] = v; zw[k >> 2] = x >> 12; }
		else {
			gm[k >> 2] = wtz_pk_max(gm[k >> 2], v);
			if constexpr((k & 3) == 1) zw[k >> 2] = wtz_bfi(0x00F000F0u, x >> 8, zw[k >> 2]);
			else if constexpr((k & 3) == 2) zw[k >> 2] = wtz_bfi(0x0F000F00u, x >> 4, zw[k >> 2]);
			else zw[k >> 2] = wtz_bfi(0xF000F000u, x, zw[k >> 2]);
		}
		if constexpr((k & 3) == 3 || k == C2 - 1){ WTZ_PIN(zw[k >> 2]); }
	});
}

/* traceback over the nibble trace of wtz_extend_shift_pk: wtz_shift_traceback<C, 64, false> with another staging step (dword q of lane l of a row holds the
 * registers 4q .. 4q+3 of the lane: run A's nibbles in bits 0..15, run B's in bits 16..31) */
template<int C2>
WTZ_D bool wtz_shift_traceback_pk(wtz_aln_t &x, uint8_t **zchunk, const int32_t *zb, uint32_t zrow, uint32_t *lds, wtz_cigar_t &cigars, const wtz_tb_score *sc){
	const int lane = (int)(threadIdx.x & 63);
	constexpr int C = 2 * C2, CQ = (C2 + 3) / 4, NL = 64;
	constexpr int NLW = (128 / C) < NL ? (128 / C) : NL;     /* lanes of a row inside the window */
	constexpr int NDW = NLW * CQ, ROWB = NLW * C;
	static_assert(NDW <= 64 && ROWB <= 128, "window geometry");
	uint32_t *S32 = lds; uint8_t *S8 = (uint8_t*)lds; uint8_t *Sd = S8 + 8192;
	int32_t i_ = x.qe, j_ = x.te; uint32_t d_ = 0;
	uint32_t run_op = 0xFFu, run_len = 0;
	int32_t n_gap_runs = 0; bool consistent = true;
	wtz_cigw_t Wr; Wr.v = &cigars; Wr.tail = 0;
	int32_t cc = 0;
	if(i_ >= 0) cc = j_ - wtz_as_global(zb)[i_];
	const int ln_off = lane / CQ, q4 = lane % CQ;
	while(i_ >= 0 && j_ >= 0){
		const int32_t i0 = i_;
		int32_t L0 = (cc < 0 ? 0 : (cc > NL * C - 1 ? NL * C - 1 : cc)) / C - NLW / 2;
		if(L0 > NL - NLW) L0 = NL - NLW;
		if(L0 < 0) L0 = 0;
		const int32_t CC0 = L0 * C;
		{
			const int32_t r = i0 - lane;
			Sd[lane] = (r >= 1) ? (uint8_t)(wtz_as_global(zb)[r] - wtz_as_global(zb)[r - 1]) : (uint8_t)0;
		}
		const int32_t cA = i0 >> 6;
		const uint8_t *chA = wtz_as_global(zchunk)[cA];
		const uint8_t *chB = cA > 0 ? wtz_as_global(zchunk)[cA - 1] : chA;
		const bool act = lane < NDW && (L0 + ln_off) < NL;
		const uint32_t doff = act ? ((uint32_t)q4 * (uint32_t)NL + (uint32_t)(L0 + ln_off)) * 4u : 0u;
		const uint32_t pos0 = (uint32_t)ln_off * (uint32_t)C + (uint32_t)q4 * 4u;
		for(int r8 = 0; r8 < 64; r8 += 8){
			uint32_t w8[8];
			#pragma unroll
			for(int u = 0; u < 8; u++){
				int32_t r = i0 - (r8 + u); if(r < 0) r = 0;
				const uint8_t *rowp = ((r >> 6) == cA ? chA : chB) + (size_t)(r & 63) * zrow;
				w8[u] = *wtz_as_global((const uint32_t*)(rowp + doff));
			}
			if(act){
				#pragma unroll
				for(int u = 0; u < 8; u++){
					const uint32_t pos = (uint32_t)(r8 + u) * 128u + pos0;
					#pragma unroll
					for(int hh = 0; hh < 2; hh++){
						/* four nibbles -> four bytes -> the walker's bytes: bits 1:0 move from H, bits 3:2 from E, bits 5:4 from F */
						uint32_t w = hh ? (w8[u] >> 16) : (w8[u] & 0xFFFFu);
						w = (w | (w << 8)) & 0x00FF00FFu; w = (w | (w << 4)) & 0x0F0F0F0Fu;
						const uint32_t a = (w >> 2) & 0x01010101u, b = (w >> 3) & 0x01010101u;
						const uint32_t v = (a << 1) | (b & (a ^ 0x01010101u)) | ((w & 0x02020202u) << 1) | ((w & 0x01010101u) << 5);
						const uint32_t p = pos + (hh ? (uint32_t)C2 : 0u);
						if constexpr((C2 & 3) == 0) S32[p >> 2] = v;
						else {
							#pragma unroll
							for(int k = 0; k < 4; k++) if(q4 * 4 + k < C2) S8[p + k] = (uint8_t)(v >> (8 * k));
						}
					}
				}
			}
		}
		__threadfence_block();
		if(lane == 0){
			while(i_ >= 0 && j_ >= 0){
				const int32_t rr = i0 - i_;
				if(rr >= 64) break;
				uint32_t zv = 0;
				if((uint32_t)cc < (uint32_t)(NL * C)){
					const int32_t t = cc - CC0;
					if((uint32_t)t >= (uint32_t)ROWB) break;
					zv = S8[rr * 128 + t];
				}
				const int32_t sft = (int32_t)Sd[rr];
				d_ = (zv >> (d_ << 1)) & 0x03;
				if(d_ == 0){ x.mat++; i_--; j_--; cc += sft - 1; }      /* x.mat counts the diagonal steps until the end */
				else if(d_ == 1){ i_--; x.ins++; cc += sft; }
				else { j_--; x.del++; cc--; }
				if(d_ == run_op) run_len++;
				else { if(run_len) wtz_cigw_push(Wr, run_op, run_len); run_op = d_; run_len = 1; if(d_) n_gap_runs++; }
			}
		}
		i_ = __builtin_amdgcn_readfirstlane(i_); j_ = __builtin_amdgcn_readfirstlane(j_); cc = __builtin_amdgcn_readfirstlane(cc);
		__threadfence_block();
	}
	if(lane == 0){
		if(run_len) wtz_cigw_push(Wr, run_op, run_len);
		if(i_ >= 0){ x.ins += i_ + 1; wtz_cigw_push(Wr, 1, (uint32_t)(i_ + 1)); n_gap_runs++; }
		if(j_ >= 0){ x.del += j_ + 1; wtz_cigw_push(Wr, 2, (uint32_t)(j_ + 1)); n_gap_runs++; }
		wtz_cigw_finish(Wr);
		wtz_cigar_reverse(cigars.a, cigars.n);
		const int32_t nd = x.mat, MXd = sc->M - sc->X;
		const long long S = (long long)x.score - sc->init - (long long)n_gap_runs * sc->O - (long long)sc->E * (x.ins + x.del) - (long long)sc->X * nd;
		if(MXd == 0 || S % MXd != 0 || S / MXd < 0 || S / MXd > nd) consistent = false;
		else { x.mat = (int32_t)(S / MXd); x.mis = nd - x.mat; }
		x.aln = x.mat + x.mis + x.ins + x.del; x.qe++; x.te++;
	}
	return __builtin_amdgcn_readfirstlane((int)consistent) != 0;
}

/* tp: LDS, 2*TWD dwords: the low-bit plane of the target in tp[0 .. TWD), the high-bit plane behind it */
template<int C2, int TWD>
WTZ_D wtz_aln_t wtz_extend_shift_pk(int32_t qlen, const wtz_seq_packed &query, int32_t tlen, const wtz_seq_packed &target, int32_t init_score,
		int32_t ql, int32_t tl, int32_t W, int32_t M, int32_t X, int32_t O, int32_t E, int32_t T, int32_t bias, int32_t NG, int32_t SH,
		uint32_t *tp, wtz_trace_t &tr, wtz_pool_t *pool, wtz_cigar_t &cigars, unsigned long long *cells, bool *ok, bool *consistent){
	const int lane = (int)(threadIdx.x & 63);
	constexpr int C = 2 * C2, CQ = (C2 + 3) / 4;
	wtz_aln_t x; memset(&x, 0, sizeof x);
	*ok = true; *consistent = true;
	if(lane == 0) cigars.n = 0;
	if(init_score < 0) init_score = 0;
	qlen = __builtin_amdgcn_readfirstlane(qlen); tlen = __builtin_amdgcn_readfirstlane(tlen); init_score = __builtin_amdgcn_readfirstlane(init_score);
	ql = __builtin_amdgcn_readfirstlane(ql); tl = __builtin_amdgcn_readfirstlane(tl); W = __builtin_amdgcn_readfirstlane(W); bias = __builtin_amdgcn_readfirstlane(bias);
	NG = __builtin_amdgcn_readfirstlane(NG); SH = __builtin_amdgcn_readfirstlane(SH);
	M = __builtin_amdgcn_readfirstlane(M); X = __builtin_amdgcn_readfirstlane(X); O = __builtin_amdgcn_readfirstlane(O); E = __builtin_amdgcn_readfirstlane(E); T = __builtin_amdgcn_readfirstlane(T);
	const uint32_t zrow = (uint32_t)CQ * 256u;
	if(!wtz_trace_prepare(tr, pool, zrow, ql, true)){ *ok = false; return x; }
	uint8_t **zchunk = tr.chunk; int32_t *zb = tr.zb;
	uint8_t *z = NULL;
	{
		const int32_t nw = (tl + 31) / 32 + 2;
		for(int32_t w = lane; w < nw; w += 64){ const uint64_t pw = wtz_pack32(target, w * 32, tl); tp[w] = wtz_even_bits(pw); tp[TWD + w] = wtz_even_bits(pw >> 1); }
	}
	__threadfence_block();
	const int32_t colrel0 = lane * C;
	const int32_t MX = M - X, Xp = X - 2 * E;
	const uint32_t MXp = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)wtz_pk2(MX, MX)), Xpp = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)wtz_pk2(Xp, Xp)), Op = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)wtz_pk2(O, O));
	/* column term of the arg-max value of register k: k*E (run A), (C2 + k)*E (run B); wave-uniform */
	uint32_t ckp[C2];
	#pragma unroll
	for(int k = 0; k < C2; k++) ckp[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)wtz_pk2(k * E, (C2 + k) * E));
	const int32_t lcE = colrel0 * E;
	/* row 0 as a row with S = 1 over a synthetic row -1 whose frame starts at column -1 (wtz_extend_shift_fr) */
	uint32_t hv[C2], ev[C2];
	#pragma unroll
	for(int p = 0; p < C2; p++){
		int32_t vh[2], ve[2];
		#pragma unroll
		for(int hh = 0; hh < 2; hh++){
			const int32_t c = colrel0 + hh * C2 + p - 1;
			const int32_t hr = (c < 0) ? init_score : init_score + O + E * (c + 1);
			vh[hh] = hr - (c - 1) * E - bias;     /* G(-1,c) */
			ve[hh] = NG - c * E - bias;           /* E~(0,c) */
		}
		hv[p] = wtz_pk2(vh[0], vh[1]); ev[p] = wtz_pk2(ve[0], ve[1]);
	}
	int32_t mx = init_score, mi = -1, mj = -1, gmax = 0, gi = -1, gj = -1;
	int32_t jbp = -1, c = 0, i;
	unsigned long long ncell = 0;
	uint32_t qw_lo = 0, qw_hi = 0, qcur = 0;
	int32_t jb_n = 0, je_n = tl; uint32_t plo_n, phi_n;
	int32_t zbv = 0; bool lost = false;
	{
		if(je_n > W + 1) je_n = W + 1;              /* row 0: c = 0 */
		if(je_n > tl) je_n = tl;
		const int32_t jj = colrel0 < tl ? colrel0 : (tl > 0 ? tl - 1 : 0);
		const int32_t w = jj >> 5; const uint32_t sh = (uint32_t)(jj & 31);
		plo_n = __builtin_amdgcn_alignbit(tp[w + 1], tp[w], sh); phi_n = __builtin_amdgcn_alignbit(tp[TWD + w + 1], tp[TWD + w], sh);
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
		uint32_t eqw;
		{
			const uint32_t nl = (qbase & 1u) - 1u, nh = ((qbase >> 1) & 1u) - 1u;       /* ~(all ones where the query bit is set) */
			const uint32_t eq = (plo_n ^ nl) & (phi_n ^ nh);                             /* bit k: column k of the lane holds the query's base */
			if constexpr(C2 == 16) eqw = eq;
			else eqw = (eq & ((1u << C2) - 1u)) | ((eq >> C2) << 16);
		}
		const int32_t s = __builtin_amdgcn_readfirstlane(jb - jbp);
		const int32_t bnd = ((jb == 0) ? init_score + O + E * i : NG) - (i + jb - 2) * E - bias;
		const int32_t SF = NG - (i + jb) * E - bias;
		const uint32_t bndp = (uint32_t)bnd << 16, SFp = wtz_pk2(SF, SF);
		uint32_t zw[CQ], gm[CQ];
		if(s == 1)      wtz_pk_row<C2, 1>(hv, ev, zw, gm, eqw, bndp, SFp, SF, O, MXp, Xpp, Op, ckp);
		else if(s == 0) wtz_pk_row<C2, 0>(hv, ev, zw, gm, eqw, bndp, SFp, SF, O, MXp, Xpp, Op, ckp);
		else            wtz_pk_row<C2, 2>(hv, ev, zw, gm, eqw, bndp, SFp, SF, O, MXp, Xpp, Op, ckp);
		const int32_t nvt = je - jb;                       /* band-relative column of the first cell beyond the band end */
		/* ---- row maximum over the cells inside the band and its FIRST arg-max (kswx.h:172) ---- */
		const int32_t vcut = __builtin_amdgcn_readfirstlane(nvt / C2), kcut = __builtin_amdgcn_readfirstlane(nvt % C2);      /* the run the band end cuts through, its first cell beyond */
		int32_t K;
		{
			uint32_t LM = gm[0];
			#pragma unroll
			for(int g = 1; g < CQ; g++) LM = wtz_pk_max(LM, gm[g]);
			if(kcut){
				uint32_t pm = 0x80008000u;
				wtz_uniform_switch<1, C2>(kcut, [&](auto kc){
					constexpr int KC = decltype(kc)::value, GF = KC / 4;
					uint32_t r = 0x80008000u;
					#pragma unroll
					for(int g = 0; g < GF; g++) r = wtz_pk_max(r, gm[g]);
					#pragma unroll
					for(int k = 4 * GF; k < KC; k++) r = wtz_pk_max(r, wtz_pk_adds(hv[k], ckp[k]));
					pm = r; WTZ_PIN_TAG(pm, KC);
				});
				const uint32_t hmask = (vcut & 1) ? 0xFFFF0000u : 0x0000FFFFu;
				const uint32_t LMc = (pm & hmask) | (LM & ~hmask);
				LM = (lane == (vcut >> 1)) ? LMc : LM;
			}
			const int32_t vA = wtz_pk_lo(LM) + lcE, vB = wtz_pk_hi(LM) + lcE;
			const int32_t kA = (colrel0 < nvt) ? (int32_t)(((uint32_t)vA << 7) + (uint32_t)(127 - 2 * lane)) : (int32_t)0x80000000;
			const int32_t kB = (colrel0 + C2 < nvt) ? (int32_t)(((uint32_t)vB << 7) + (uint32_t)(126 - 2 * lane)) : (int32_t)0x80000000;
			K = wtz_wave_max_i32(kA > kB ? kA : kB);
		}
		ncell += (unsigned long long)(je - jb);
		int32_t imax = 0, mj2 = -1;
		{
			const int32_t Hm = (K >> 7) + bias + (i + jb) * E;
			if(Hm > 0 && Hm > SH){
				imax = Hm;
				const int32_t vidx = 127 - (K & 127), Ls = vidx >> 1, hs = vidx & 1;
				const int32_t T16 = (K >> 7) - Ls * C * E;                 /* the winning run's value of h + column*E */
				const int32_t klim = (vidx == vcut) ? kcut : C2;           /* its cells inside the band: k < klim */
				/* the first register k < klim of lane Ls whose half hs holds T16.  In every lane at once (round 6: the first form read the group maxima and the four
				 * candidates of a group into scalar registers and compared them there - a hundred scalar instructions per row, more than the row's own share of a
				 * lone wavefront's time): a bit per group maximum that equals T16, the first such group among those entirely inside the band (else the group the
				 * band end cuts), a bit per register of that group, and only the two masks of lane Ls go to the scalar unit */
				const uint32_t Tv = (uint32_t)T16 & 0xFFFFu;
				const int32_t nfull = klim >> 2;                           /* groups 0 .. nfull-1 lie entirely inside the band */
				int32_t kf = -1;
				auto find = [&](auto hc){
					constexpr int H = decltype(hc)::value;
					uint32_t ag = 0;
					#pragma unroll
					for(int g = CQ - 1; g >= 0; g--) ag = wtz_pk_eqacc<H>(ag, gm[g], Tv);            /* bit g: group g's maximum is T16 */
					const uint32_t sg = (uint32_t)__builtin_amdgcn_readlane((int32_t)ag, Ls) & ((1u << nfull) - 1u);
					const int32_t gsel = sg ? (int32_t)__builtin_ctz(sg) : nfull;
					if(gsel < CQ){
						uint32_t ae = 0;
						wtz_uniform_switch<0, CQ>(gsel, [&](auto gc){
							constexpr int g = decltype(gc)::value;
							uint32_t a = 0;
							#pragma unroll
							for(int k = 4 * g + 3; k >= 4 * g; k--) a = (k < C2) ? wtz_pk_eqacc<H>(a, wtz_pk_adds(hv[k < C2 ? k : 0], ckp[k < C2 ? k : 0]), Tv) : a + a;      /* bit k - 4g */
							ae = a; WTZ_PIN_TAG(ae, g);
						});
						const int32_t room = klim - 4 * gsel;                 /* registers of the group inside the band */
						const uint32_t se = (uint32_t)__builtin_amdgcn_readlane((int32_t)ae, Ls) & (room >= 4 ? 0xFu : ((1u << (room > 0 ? room : 0)) - 1u));
						if(se) kf = 4 * gsel + (int32_t)__builtin_ctz(se);
					}
				};
				if(hs) find(wtz_ic<1>{}); else find(wtz_ic<0>{});
				if(kf < 0){ lost = true; kf = 0; }
				mj2 = jb + Ls * C + hs * C2 + kf;
			}
		}
		zbv = (lane == (i & 63)) ? jb : zbv;
		if((i & 63) == 63) wtz_as_global(zb)[(i & ~63) + lane] = zbv;
		/* ---- H(i, je-1) for the target-end rule ---- */
		if(je == tlen){
			const int32_t idx = nvt - 1;
			const int32_t vr = __builtin_amdgcn_readfirstlane(idx / C2), kl = __builtin_amdgcn_readfirstlane(idx % C2);
			uint32_t hsel = 0;
			wtz_uniform_switch<0, C2>(kl, [&](auto kc){ constexpr int k = decltype(kc)::value; hsel = hv[k]; WTZ_PIN_TAG(hsel, k); });
			const uint32_t hw = (uint32_t)__builtin_amdgcn_readlane((int32_t)hsel, vr >> 1);
			const int32_t h1 = ((vr & 1) ? wtz_pk_hi(hw) : wtz_pk_lo(hw)) + bias + (i + je - 1) * E;      /* H(i, je-1) */
			if(h1 > SH && gmax < h1){ gmax = h1; gi = i; gj = je - 1; }
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
			if(jb_n != jb){
				const int32_t j0n = jb_n + colrel0;
				const int32_t jj = j0n < tl ? j0n : (tl > 0 ? tl - 1 : 0);
				const int32_t w = jj >> 5; const uint32_t sh = (uint32_t)(jj & 31);
				plo_n = __builtin_amdgcn_alignbit(tp[w + 1], tp[w], sh); phi_n = __builtin_amdgcn_alignbit(tp[TWD + w + 1], tp[TWD + w], sh);
			}
			/* ---- the slots the NEXT row reads beyond this row's band end (wtz_extend_shift_fr): H(i, je), E(i+1, je), E(i+1, je+1) = -10000 ---- */
			if(je_n > je){
				const int32_t SG = NG - (i + je) * E - bias, SE1 = NG - (i + 1 + je) * E - bias, SE2 = NG - (i + 2 + je) * E - bias;
				const uint32_t SGp = wtz_pk2(SG, SG), SE1p = wtz_pk2(SE1, SE1), SE2p = wtz_pk2(SE2, SE2);
				const int32_t Lb = vcut >> 1;
				const uint32_t hm1 = (vcut & 1) ? 0xFFFF0000u : 0x0000FFFFu;
				if(Lb < 64){
					wtz_uniform_switch<0, C2>(kcut, [&](auto kc){
						constexpr int k = decltype(kc)::value;
						const bool here = lane == Lb;
						hv[k] = here ? ((SGp & hm1) | (hv[k] & ~hm1)) : hv[k];
						ev[k] = here ? ((SE1p & hm1) | (ev[k] & ~hm1)) : ev[k];
						if constexpr(k + 1 < C2) ev[k + 1] = here ? ((SE2p & hm1) | (ev[k + 1] & ~hm1)) : ev[k + 1];
						else {
							/* the first cell of the next run: run B of the same lane, or run A of the next lane */
							const int32_t L2 = (vcut + 1) >> 1;
							ev[0] = (lane == L2) ? ((SE2p & ~hm1) | (ev[0] & hm1)) : ev[0];
						}
						WTZ_PIN_TAG(hv[k], k);
					});
				}
			}
		}
		if(colrel0 < nvt){
			WTZ_GLOBAL_AS uint32_t *zr = wtz_as_global((uint32_t*)(z + (size_t)(i & 63) * zrow) + lane);
			#pragma unroll
			for(int q4 = 0; q4 < CQ; q4++) zr[(size_t)q4 * 64] = zw[q4];
		}
		if(stop) break;
	}
	if(cells && lane == 0) *cells += ncell;
	if(!*ok) return x;
	{   /* band starts of the rows of the last, incomplete block of 64 */
		const int32_t last = i < ql ? i : ql - 1;
		if(last >= 0 && (last & 63) != 63 && lane <= (last & 63)) wtz_as_global(zb)[(last & ~63) + lane] = zbv;
	}
	if(gmax > 0 && gmax >= mx + T){ x.score = gmax; x.qe = gi; x.te = gj; }
	else { x.score = mx; x.qe = mi; x.te = mj; }
	__threadfence_block();
	const wtz_tb_score sc = { M, X, O, E, init_score };
	if(!wtz_shift_traceback_pk<C2>(x, zchunk, zb, zrow, tp, cigars, &sc)) *ok = false, *consistent = false;
	if(__builtin_amdgcn_readfirstlane((int)lost)) *ok = false, *consistent = false;      /* the arg-max search came back empty: never observed; the job stays open for the 32-bit form */
	return wtz_bcast_aln(x);
}

/* the 16-bit window of a job: false = it does not fit (or the scores are outside what the packed row assumes).
 * (a) the reference's own numbers: [ -10000 + the deepest fall of a row's lowest value, init + the highest score ] within 16 bits: *ng = -10000, no sum saturates,
 *     every value is the reference's minus the bias.
 * (b) a job whose init_score alone pushes (a) beyond 16 bits (the right extension of a long overlap): the -10000 family stands in for "minus infinity", and it may
 *     stand HIGHER as long as it still loses every comparison the reference's loses.  A value of the DP is the best path from a source; the sources are the
 *     real ones (row -1, column -1) and the -10000 entries.  Every path from a real source is worth at least
 *         Rlow = init - |O| - (tl+1)|E| - (ql+1)*max(|X|, |O|+|E|) - (ql+2)*|O|
 *     (the horizontal movement of a path, its source's included, costs at most (tl+1)|E| and one opening per row; each row-advancing step, the left boundary's
 *     included, at most max(|X|, |O|+|E|)), every path from a -10000 source at most -10000 + M*min(ql,tl) (< 0 when M*min(ql,tl) < 10000: in the reference such
 *     a value never passes a `> 0` test).  With the family at
 *         NG = Rlow - M*min(ql,tl) - 64 (> -10000)
 *     every state that has a real path takes the reference's value (the real path wins in both worlds); a state without one holds SOME value <= NG + M*min(ql,tl)
 *     (sums saturate at the bottom of the 16 bits, which only ever replaces a family value by another one not above NG) that loses against every real value
 *     like the reference's.  The decisions on the path of the result, the row maxima and their arg-max columns compare real values with each other or a real
 *     winner with a family loser: they are the reference's.  The two places that test a value against zero (row maximum, target-end cell) take a family value
 *     (<= *sh = NG + M*min(ql,tl)) for what it is in the reference: negative. */
WTZ_D bool wtz_pk_window(const wtz_params_t *Pm, int32_t init_score, int32_t ql, int32_t tl, int32_t *bias, int32_t *ng, int32_t *sh){
	const long long M = Pm->M, X = Pm->X, O = Pm->O, E = Pm->E;
	if(E > 0 || E < -255 || X > 0 || M < 0 || O > 0 || M == X || M - X > 4096) return false;
	const long long aE = -E, aO = -O, aX = -X, Xp = X - 2 * E, Xm = Xp < 0 ? Xp : 0;
	const long long mn = ql < tl ? ql : tl;
	const long long hi = (long long)init_score + M * mn + ((long long)ql + tl + 4) * aE + (M - X) + 64;
	long long NG = -10000;
	long long lo = NG + ((long long)ql + 3) * Xm + O - 33 * aE - 64;
	*sh = -(1 << 30);
	if(hi - lo > 65000){
		if(M * mn >= 10000) return false;
		const long long step = aX > aO + aE ? aX : aO + aE;
		const long long Rlow = (long long)init_score - aO - ((long long)tl + 1) * aE - ((long long)ql + 1) * step - ((long long)ql + 2) * aO;
		NG = Rlow - M * mn - 64;
		if(NG <= -10000) return false;
		lo = NG - 33 * aE - 64;
		if(hi - lo > 65000) return false;
		*sh = (int32_t)(NG + M * mn);
	}
	*bias = (int32_t)((hi + lo) / 2); *ng = (int32_t)NG;
	return true;
}

/* Does a job of this geometry fit the packed form for EVERY init_score in [i_lo, i_hi]?  (The right extension's init_score is the score of everything before it;
 * when a stage's items are dealt to the two forms all of it is known but the left extension's own gain: a range.)  Window (a) holds up to an init_score Ia,
 * window (b) from Ib on. */
WTZ_D bool wtz_pk_window_range(const wtz_params_t *Pm, int32_t ql, int32_t tl, long long i_lo, long long i_hi){
	const long long M = Pm->M, X = Pm->X, O = Pm->O, E = Pm->E;
	if(E > 0 || E < -255 || X > 0 || M < 0 || O > 0 || M == X || M - X > 4096) return false;
	if(i_lo < 0) i_lo = 0;
	if(i_hi < 0) i_hi = 0;
	const long long aE = -E, aO = -O, aX = -X, Xp = X - 2 * E, Xm = Xp < 0 ? Xp : 0;
	const long long mn = ql < tl ? ql : tl;
	const long long top = M * mn + ((long long)ql + tl + 4) * aE + (M - X) + 64;                       /* hi - init */
	const long long lo_a = -10000 + ((long long)ql + 3) * Xm + O - 33 * aE - 64;
	const long long Ia = 65000 + lo_a - top;                                                            /* (a) fits up to this init_score */
	if(i_hi <= Ia) return true;
	if(M * mn >= 10000) return false;
	const long long step = aX > aO + aE ? aX : aO + aE;
	const long long fall = aO + ((long long)tl + 1) * aE + ((long long)ql + 1) * step + ((long long)ql + 2) * aO;      /* init - Rlow */
	const long long Ib = -10000 + 64 + M * mn + fall;                                                   /* (b)'s family stands above -10000 beyond this init_score */
	const long long width_b = top + fall + M * mn + 64 + 33 * aE + 64;
	if(width_b > 65000) return false;
	return i_lo > Ib || Ib < Ia;          /* all of the range in (b), or the two windows overlap */
}

/* one K-sw3 job on the calling wavefront in the packed form; false = declined (outside the window or the envelope of wtz_extjob_run_fr: the job stays open).
 * stb: TW 64-bit words of LDS of this wave. */
template<int TW>
WTZ_D bool wtz_extjob_run_pk(wtz_extjob_t *job, const wtz_params_t *Pm, wtz_pool_t *pool, wtz_pool_t *tpool, uint64_t *stb){
	if(!job->valid || job->done) return true;
	const int lane = (int)(threadIdx.x & 63);
	if(job->qlen <= 0 || job->tlen <= 0) return false;
	const int32_t init_score = job->init_score < 0 ? 0 : job->init_score;
	int32_t W = job->W, ql, tl, n_col;
	wtz_ext_geometry(job->qlen, job->tlen, init_score, W, Pm->M, Pm->O, Pm->O, Pm->E, Pm->T, ql, tl, n_col);
	const int32_t Cw = (n_col + 63) / 64;
	if(Cw > 32 || (tl + 31) / 32 + 3 > TW || (ql + 63) / 64 > WTZ_TRACE_MAXCHUNK) return false;
	int32_t bias = 0, ng = -10000, sh = 0;
	if(!wtz_pk_window(Pm, init_score, ql, tl, &bias, &ng, &sh)) return false;
	wtz_trace_t tr; tr.chunk = NULL; tr.zb = NULL; tr.n_chunk = 0; tr.zrow = 0; tr.cap_rows = 0;
	wtz_cigar_t cg; cg.a = NULL; cg.n = cg.cap = 0; cg.pool = pool; cg.bad = 0;
	if(lane == 0) cg.init(pool, (uint32_t)ql / 2u + 16u);
	unsigned long long cells = 0; bool ok = true, consistent = true;
	wtz_aln_t x;
#define WTZ_EXTPK_CASE(CM) x = wtz_extend_shift_pk<CM, TW>(job->qlen, job->q, job->tlen, job->t, job->init_score, ql, tl, W, Pm->M, Pm->X, Pm->O, Pm->E, Pm->T, bias, ng, sh, (uint32_t*)stb, tr, tpool, cg, &cells, &ok, &consistent)
	/* one instantiation per register of a lane (two columns): a job's band is at most two columns per lane narrower than its class */
	if(Cw <= 4) WTZ_EXTPK_CASE(2);
	else if(Cw <= 6) WTZ_EXTPK_CASE(3);
	else if(Cw <= 8) WTZ_EXTPK_CASE(4);
	else if(Cw <= 10) WTZ_EXTPK_CASE(5);
	else if(Cw <= 12) WTZ_EXTPK_CASE(6);
	else if(Cw <= 14) WTZ_EXTPK_CASE(7);
	else if(Cw <= 16) WTZ_EXTPK_CASE(8);
	else if(Cw <= 18) WTZ_EXTPK_CASE(9);
	else if(Cw <= 20) WTZ_EXTPK_CASE(10);
	else if(Cw <= 22) WTZ_EXTPK_CASE(11);
	else if(Cw <= 24) WTZ_EXTPK_CASE(12);
	else if(Cw <= 26) WTZ_EXTPK_CASE(13);
	else if(Cw <= 28) WTZ_EXTPK_CASE(14);
	else if(Cw <= 30) WTZ_EXTPK_CASE(15);
	else WTZ_EXTPK_CASE(16);
#undef WTZ_EXTPK_CASE
	if(!consistent) return false;
	if(lane == 0){ job->x = x; job->cigar = cg.a; job->cigar_len = cg.n; job->bad = (!ok || cg.bad); job->cells = cells; job->done = 7; }
	return true;
}

/* one wavefront per job */
template<int TW>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WTZ_OCC_EXTPK, 8))) wtz_kernel_extjobs_pk(wtz_extjob_t *jobs, const uint32_t *order, uint32_t n, const wtz_params_t *Pm, wtz_pool_t *pool, wtz_pool_t *tpool){
	__shared__ uint64_t stb[TW];
	const uint32_t b = blockIdx.x;
	if(b >= n) return;
	(void)wtz_extjob_run_pk<TW>(&jobs[order ? order[b] : b], Pm, pool, tpool, stb);
}

#endif /* __HIPCC__ */
#endif
