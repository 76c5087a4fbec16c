/*
 * K-sw3 (kswx_extend_align_shift_core, /root/reference/kswx.h:101-232) in the anti-diagonal frame on FOUR wavefronts per job (round 6).
 *
 * Why: a job's rows are sequential and a launch ends when its longest job does.  The kernel trace of a configs[2] step
 * (profiles/r06_ksw3_split_in_step_kernel_trace.txt) shows it plainly: the 10 000 widest-band jobs of a side take 17 ms on one wavefront each, the 43 000 others
 * 4.7 ms - the stage is the critical path of a handful of jobs of 4 000-8 000 rows at ~2.2 us per row of 28 columns per lane.  A lone wavefront issues one
 * instruction per ~5 cycles at best (its SIMD's turn in the CU's issue rotation), so the only way to make ONE job's row shorter is to issue it from more
 * SIMDs: 256 lanes of C <= 8 columns each, every wave a quarter of the cells (wtz_fr_row_x: the cell body of wtz_sw_frame.h), two workgroup barriers per row:
 *   barrier 1  after each wave's in-wave prefix maximum of the F aggregates: the totals of the waves to the left complete a lane's F carry-in;
 *   barrier 2  after the cells: every wave's arg-max key, the H / E values its neighbours' edge lanes will shift in, and H(i, je-1) on rows that touch the
 *              target's end; behind it every thread derives the same row maximum, band centre and stop decision.
 * The values that cross a wave boundary replace the `old` operand of the edge lane's DPP move (wave_shr:1 into lane 0, wave_shl:1 into lane 63).  The three slots
 * beyond the band end that a growing band reads (wtz_sw_frame.h) are set by the thread that owns them; where such a slot is a wave's edge value, the
 * neighbour that read it before the band moved patches its copy with the same scalar (wtz_fr_edge_fix).
 * The round-4 four-wave kernel (wtz_extend_shift_mw, wtz_sw_wave.h) had the round-4 row body - 1.84 x the instructions per row - and was off since round 5.
 * Trace layout (row, k/4, thread, k%4) with 256 threads per row; wtz_shift_traceback<C, 256, false> on wave 0.
 */
#ifndef WTZ_SW_FRAME_MW_H
#define WTZ_SW_FRAME_MW_H

#include "wtz_sw_frame.h"

#ifdef __HIPCC__

/* what the waves of a job exchange (LDS, one per workgroup) */
typedef struct {
	wtz_i4 wagg, wkey, edge[4];         /* edge[w]: v[0] = hv[C-1] of lane 63, v[1] = hv[0], v[2] = ev[0], v[3] = ev[1] of lane 0 */
	int32_t h1, flag; unsigned long long zbase, chunk_tab, zb_tab;
} wtz_frmw_shared_t;

/* one row of the frame DP for the C cells of one thread of a multi-wave job: wtz_fr_row with the three values an edge lane takes from the neighbouring wave
 * (or the band boundary) and the F carry of the waves to the left, which arrives through LDS BETWEEN the two passes */
template<int C, int S>
WTZ_D int32_t wtz_fr_row_x_pass1(int32_t (&hv)[C], const uint32_t eq_lo, const uint32_t eq_hi, const int32_t h_prev, const int32_t h0_next, const int32_t MX, const int32_t Xp){
	int32_t agg = (int32_t)0x80000000;
	if constexpr(S == 0){
		const int32_t prv = wtz_dpp_wave_shr1(h_prev, hv[C - 1]);      /* lane 0 keeps h_prev: the last column of the wave to the left, or the band boundary */
		wtz_static_for<0, C>([&](auto kc){
			constexpr int k = C - 1 - decltype(kc)::value;
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
		const int32_t nh0 = wtz_dpp_wave_shl1(h0_next, hv[0]);        /* lane 63 keeps h0_next: the first column of the wave to the right */
		wtz_static_for<0, C>([&](auto kc){
			constexpr int k = decltype(kc)::value;
			const int32_t b = (int32_t)(((k < 16 ? eq_lo : eq_hi) >> (2 * (k & 15))) & 1u);
			const int32_t src = (k == C - 1) ? nh0 : hv[k == C - 1 ? k : k + 1];
			const int32_t m = wtz_mad24(b, MX, src) + Xp;
			hv[k] = m; agg = m > agg ? m : agg;
		});
	}
	return agg;
}
template<int C, int S>
WTZ_D void wtz_fr_row_x_pass2(int32_t (&hv)[C], int32_t (&ev)[C], uint32_t (&zw)[(C + 3) / 4], int32_t f, const int32_t e0_next, const int32_t e1_next,
		const int32_t O, const int32_t (&ck)[C], int32_t &lkey){
	int32_t ne0 = 0, ne1 = 0;
	if constexpr(S >= 1) ne0 = wtz_dpp_wave_shl1(e0_next, ev[0]);
	if constexpr(S == 2) ne1 = wtz_dpp_wave_shl1(e1_next, ev[C > 1 ? 1 : 0]);
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
		uint32_t d = (uint32_t)(m - e) >> 31;
		d = __builtin_amdgcn_alignbit(d, (uint32_t)(h0 - f), 31);
		const int32_t h = h0 > f ? h0 : f;
		const int32_t t = m + O;
		d = __builtin_amdgcn_alignbit(d, (uint32_t)(t - e), 31);
		const int32_t en = e > t ? e : t;
		d = __builtin_amdgcn_alignbit(d, (uint32_t)(t - f), 31);
		f = f > t ? f : t;
		hv[k] = h; ev[k] = en;
		const int32_t kk = (int32_t)(((uint32_t)h << 11) + (uint32_t)ck[k]);
		key = kk > key ? kk : key;
		zw[k >> 2] |= d << (8 * (k & 3));
		if constexpr((k & 3) == 3 || k == C - 1){ WTZ_PIN(zw[k >> 2]); }
	});
	lkey = key;
}

/* a value read from a neighbouring wave's edge before the band moved: band-relative column p of the row just computed.  If that column is one of the slots the
 * owner resets for a growing band (wtz_sw_frame.h: H at nvt, E at nvt and nvt + 1), the copy gets the same scalar */
WTZ_D int32_t wtz_fr_edge_fix_h(int32_t v, int32_t p, int32_t nvt, int32_t SG){ return p == nvt ? SG : v; }
WTZ_D int32_t wtz_fr_edge_fix_e(int32_t v, int32_t p, int32_t nvt, int32_t SE1, int32_t SE2){ return p == nvt ? SE1 : (p == nvt + 1 ? SE2 : v); }

template<int C, int NW>
WTZ_D wtz_aln_t wtz_extend_shift_frmw(int32_t qlen, const wtz_seq_packed &query, int32_t tlen, const wtz_seq_packed &target, int32_t init_score,
		int32_t ql, int32_t tl, int32_t W, int32_t M, int32_t X, int32_t O, int32_t E, int32_t T,
		uint64_t *tb, wtz_frmw_shared_t *sh_generic, wtz_pool_t *pool, wtz_cigar_t &cigars, unsigned long long *cells, bool *ok, bool *consistent){
	static_assert(C >= 2 && NW == 4, "lane block of at least two columns, four waves");
	WTZ_LDS_AS wtz_frmw_shared_t *sh = wtz_as_lds(sh_generic);
	const int tid = (int)threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
	constexpr int C4 = (C + 3) / 4, NL = 64 * NW;
	wtz_aln_t x; memset(&x, 0, sizeof x);
	*ok = true; *consistent = true;
	if(tid == 0) cigars.n = 0;
	if(init_score < 0) init_score = 0;
	qlen = __builtin_amdgcn_readfirstlane(qlen); tlen = __builtin_amdgcn_readfirstlane(tlen); init_score = __builtin_amdgcn_readfirstlane(init_score);
	ql = __builtin_amdgcn_readfirstlane(ql); tl = __builtin_amdgcn_readfirstlane(tl); W = __builtin_amdgcn_readfirstlane(W);
	M = __builtin_amdgcn_readfirstlane(M); X = __builtin_amdgcn_readfirstlane(X); O = __builtin_amdgcn_readfirstlane(O); E = __builtin_amdgcn_readfirstlane(E); T = __builtin_amdgcn_readfirstlane(T);
	const uint32_t zrow = (uint32_t)C4 * 4u * (uint32_t)NL;
	if(tid == 0){
		sh->chunk_tab = (unsigned long long)(uintptr_t)wtz_pool_alloc(pool, (size_t)WTZ_TRACE_MAXCHUNK * 8);
		sh->zb_tab = (unsigned long long)(uintptr_t)wtz_pool_alloc(pool, (size_t)(ql + 66) * 4);      /* band starts are written 64 at a time */
		sh->flag = 0;
	}
	{
		const int32_t nw = (tl + 31) / 32 + 1;
		for(int32_t w = tid; w < nw; w += NL) tb[w] = wtz_pack32(target, w * 32, tl);
	}
	const int32_t colrel0 = tid * C;
	const int32_t MX = M - X, Xp = X - 2 * E;
	int32_t ck[C];
	#pragma unroll
	for(int k = 0; k < C; k++) ck[k] = __builtin_amdgcn_readfirstlane(k * (2048 * E - 1));
	/* row 0 as a row with S = 1 over a synthetic row -1 whose frame starts at column -1 (wtz_extend_shift_fr): slot p of thread t = column t*C + p - 1 */
	int32_t hv[C], ev[C];
	#pragma unroll
	for(int p = 0; p < C; p++){
		const int32_t c0 = colrel0 + p - 1;
		const int32_t hr = (c0 < 0) ? init_score : init_score + O + E * (c0 + 1);
		hv[p] = hr - (c0 - 1) * E;
		ev[p] = -10000 - c0 * E;
	}
	/* the edges of the synthetic row, for row 0's S = 1 body (only the E of the next wave's first slot is read) */
	if(lane == 0){ sh->edge[wid].v[1] = hv[0]; sh->edge[wid].v[2] = ev[0]; sh->edge[wid].v[3] = ev[C > 1 ? 1 : 0]; }
	if(lane == 63) sh->edge[wid].v[0] = hv[C - 1];
	__syncthreads();
	uint8_t **zchunk = (uint8_t**)(uintptr_t)sh->chunk_tab; int32_t *zb = (int32_t*)(uintptr_t)sh->zb_tab;
	if(zchunk == NULL || zb == NULL){ *ok = false; return x; }
	uint8_t *z = NULL;
	int32_t mx = init_score, mi = -1, mj = -1, gmax = 0, gi = -1, gj = -1;
	int32_t jbp = -1, c = 0, i;
	unsigned long long ncell = 0;
	uint32_t qw_lo = 0, qw_hi = 0, qcur = 0;
	int32_t jb_n = 0, je_n = tl; uint64_t tbits_n;
	int32_t koff = (colrel0 * E) * 2048 + 2047 - colrel0;
	const int32_t E2048 = E * 2048;
	int32_t zbv = 0;
	/* the reset of the previous row as the neighbours have to see it: nvt of that row (-1: no reset was due) and its three sentinels */
	int32_t fix_nvt = -1, fix_SG = 0, fix_SE1 = 0, fix_SE2 = 0;
	{
		if(je_n > W + 1) je_n = W + 1;
		if(je_n > tl) je_n = tl;
		const int32_t jj = colrel0 < tl ? colrel0 : (tl > 0 ? tl - 1 : 0);
		const int32_t w = jj >> 5, shb = (jj & 31) * 2;
		const uint64_t w0 = tb[w], w1 = tb[w + 1];
		tbits_n = shb ? ((w0 >> shb) | (w1 << (64 - shb))) : w0;
	}
	__builtin_amdgcn_s_waitcnt(0x0F70);
	bool failed = false;
	for(i = 0; i < ql; i++){
		if((i & 63) == 0){
			if(tid == 0){ uint8_t *p = (uint8_t*)wtz_pool_alloc(pool, (size_t)zrow * 64); wtz_as_global(zchunk)[(uint32_t)i >> 6] = p; sh->zbase = (unsigned long long)(uintptr_t)p; }
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
		const int32_t bnd = ((jb == 0) ? init_score + O + E * i : -10000) - (i + jb - 2) * E;
		const int32_t SF = -10000 - (i + jb) * E;
		/* the neighbouring waves' edge values of the previous row (published in front of its barrier 2), patched where the owner has reset them since */
		int32_t h_prev = bnd, h0_next = SF, e0_next = SF, e1_next = SF;
		{
			const int wl = wid > 0 ? wid - 1 : 0, wr = wid + 1 < NW ? wid + 1 : wid;
			const wtz_i4 el = sh->edge[wl], er = sh->edge[wr];
			if(wid > 0) h_prev = wtz_fr_edge_fix_h(el.v[0], wid * 64 * C - 1, fix_nvt, fix_SG);
			if(wid + 1 < NW){
				const int32_t p0 = (wid + 1) * 64 * C;
				h0_next = wtz_fr_edge_fix_h(er.v[1], p0, fix_nvt, fix_SG);
				e0_next = wtz_fr_edge_fix_e(er.v[2], p0, fix_nvt, fix_SE1, fix_SE2);
				e1_next = wtz_fr_edge_fix_e(er.v[3], p0 + 1, fix_nvt, fix_SE1, fix_SE2);
			}
		}
		/* ---- pass 1 + the F carry-in across the waves ---- */
		int32_t agg;
		if(s == 1)      agg = wtz_fr_row_x_pass1<C, 1>(hv, eq_lo, eq_hi, h_prev, h0_next, MX, Xp);
		else if(s == 0) agg = wtz_fr_row_x_pass1<C, 0>(hv, eq_lo, eq_hi, h_prev, h0_next, MX, Xp);
		else            agg = wtz_fr_row_x_pass1<C, 2>(hv, eq_lo, eq_hi, h_prev, h0_next, MX, Xp);
		int32_t f;
		{
			const int32_t mn = (int32_t)0x80000000;
			int32_t xs = agg + O, t;
			t = wtz_dpp_mov<0x111, 0xF>(mn, xs); xs = xs > t ? xs : t;
			t = wtz_dpp_mov<0x112, 0xF>(mn, xs); xs = xs > t ? xs : t;
			t = wtz_dpp_mov<0x114, 0xF>(mn, xs); xs = xs > t ? xs : t;
			t = wtz_dpp_mov<0x118, 0xF>(mn, xs); xs = xs > t ? xs : t;
			t = wtz_dpp_mov<0x142, 0xA>(mn, xs); xs = xs > t ? xs : t;
			t = wtz_dpp_mov<0x143, 0xC>(mn, xs); xs = xs > t ? xs : t;
			if(lane == 63) sh->wagg.v[wid] = xs;
			int32_t pm = wtz_dpp_mov<0x138, 0xF>(SF, xs);                     /* exclusive: lane 0 starts from the floor */
			wtz_mw_barrier();                                                   /* ---- barrier 1 ---- */
			const wtz_i4 wa = sh->wagg;
			#pragma unroll
			for(int w2 = 0; w2 + 1 < NW; w2++){ const int32_t v = wa.v[w2]; pm = (w2 < wid && v > pm) ? v : pm; }
			f = pm > SF ? pm : SF;
		}
		if((i & 63) == 0){
			const unsigned long long za = sh->zbase;
			const uint32_t zlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)za), zhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(za >> 32));
			z = (uint8_t*)(uintptr_t)(((unsigned long long)zhi << 32) | zlo);
			if((zlo | zhi) == 0){ failed = true; break; }                       /* the same value in every thread: all leave together */
		}
		uint32_t zw[C4]; int32_t key;
		if(s == 1)      wtz_fr_row_x_pass2<C, 1>(hv, ev, zw, f, e0_next, e1_next, O, ck, key);
		else if(s == 0) wtz_fr_row_x_pass2<C, 0>(hv, ev, zw, f, e0_next, e1_next, O, ck, key);
		else            wtz_fr_row_x_pass2<C, 2>(hv, ev, zw, f, e0_next, e1_next, O, ck, key);
		const int32_t nvt = je - jb;
		{
			const bool part = colrel0 < nvt && colrel0 + C > nvt;
			const int32_t best_col = 2047 - ((key + koff) & 2047);
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
		if(lane == 0){ sh->wkey.v[wid] = key; sh->edge[wid].v[1] = hv[0]; sh->edge[wid].v[2] = ev[0]; sh->edge[wid].v[3] = ev[C > 1 ? 1 : 0]; }
		if(lane == 63) sh->edge[wid].v[0] = hv[C - 1];
		if(je == tlen){
			const int32_t idx = nvt - 1, gl = idx / C, kl = idx - gl * C;
			if(tid == gl){
				int32_t hsel = hv[0];
				#pragma unroll
				for(int k = 1; k < C; k++) hsel = (kl == k) ? hv[k] : hsel;
				sh->h1 = hsel;
			}
		}
		if(wid == 0){
			zbv = (lane == (i & 63)) ? jb : zbv;
			if((i & 63) == 63) wtz_as_global(zb)[(i & ~63) + lane] = zbv;
		}
#ifndef WTZ_EXP_NOTRACE
		if(colrel0 < nvt){
			WTZ_GLOBAL_AS uint32_t *zr = wtz_as_global((uint32_t*)(z + (size_t)(i & 63) * zrow) + tid);
			#pragma unroll
			for(int q4 = 0; q4 < C4; q4++) zr[(size_t)q4 * NL] = zw[q4];
		}
#endif
		wtz_mw_barrier();                                                       /* ---- barrier 2 ---- */
		const wtz_i4 wkk = sh->wkey;
		#pragma unroll
		for(int w2 = 0; w2 < NW; w2++){ const int32_t v = wkk.v[w2]; key = v > key ? v : key; }
		key = __builtin_amdgcn_readfirstlane(key);
		int32_t imax = 0, mj2 = -1;
		if((key >> 11) > 0){ imax = key >> 11; mj2 = jb + (2047 - (key & 2047)); }
		if(je == tlen){
			const int32_t h1 = __builtin_amdgcn_readfirstlane(sh->h1) + (i + je - 1) * E;      /* H(i, je-1) out of its frame image */
			if(gmax < h1){ gmax = h1; gi = i; gj = je - 1; }
		}
		if(i + 1 == qlen && gmax < imax){ gmax = imax; gi = i; gj = mj2; }
		jbp = jb;
		if(imax > mx){ mx = imax; mi = i; mj = mj2; }
		else if(imax <= 0) break;
		c++; if(c < mj2) c++; else if(c > mj2) c--;
		jb_n = 0; je_n = tl;
		if(jb_n < c - W) jb_n = c - W;
		if(je_n > c + W + 1) je_n = c + W + 1;
		if(je_n > tl) je_n = tl;
		koff += (1 + jb_n - jb) * E2048;
		if(jb_n != jb){
			const int32_t j0n = jb_n + colrel0;
			const int32_t jj = j0n < tl ? j0n : (tl > 0 ? tl - 1 : 0);
			const int32_t w = jj >> 5, shb = (jj & 31) * 2;
			const uint64_t w0 = tb[w], w1 = tb[w + 1];
			tbits_n = shb ? ((w0 >> shb) | (w1 << (64 - shb))) : w0;
		}
		fix_nvt = -1;
		if(je_n > je){
			/* the slots a growing band reads beyond this row's end (wtz_sw_frame.h); the owner sets them, the neighbouring wave patches its copy of an edge value */
			const int32_t SG = -10000 - (i + je) * E, SE1 = -10000 - (i + 1 + je) * E, SE2 = -10000 - (i + 2 + je) * E;
			fix_nvt = nvt; fix_SG = SG; fix_SE1 = SE1; fix_SE2 = SE2;
			const int32_t Lb = nvt / C, kb = nvt - Lb * C;
			if(Lb < NL && __builtin_amdgcn_readfirstlane(Lb >> 6) == wid){
				wtz_uniform_switch<0, C>(__builtin_amdgcn_readfirstlane(kb), [&](auto kc){
					constexpr int k = decltype(kc)::value;
					hv[k] = (tid == Lb) ? SG : hv[k];
					ev[k] = (tid == Lb) ? SE1 : ev[k];
					if constexpr(k + 1 < C) ev[k + 1] = (tid == Lb) ? SE2 : ev[k + 1];
					WTZ_PIN_TAG(hv[k], k);
				});
			}
			if(kb == C - 1 && Lb + 1 < NL && __builtin_amdgcn_readfirstlane((Lb + 1) >> 6) == wid) ev[0] = (tid == Lb + 1) ? SE2 : ev[0];      /* E at nvt + 1 is the next thread's first slot */
		}
	}
	if(cells && tid == 0) *cells += ncell;
	if(failed){ *ok = false; return x; }
	if(wid == 0){
		const int32_t last = i < ql ? i : ql - 1;
		if(last >= 0 && (last & 63) != 63 && lane <= (last & 63)) wtz_as_global(zb)[(last & ~63) + lane] = zbv;
	}
	__syncthreads();                /* every wave's trace is visible to wave 0; the target words in LDS are dead */
	if(gmax > 0 && gmax >= mx + T){ x.score = gmax; x.qe = gi; x.te = gj; }
	else { x.score = mx; x.qe = mi; x.te = mj; }
	if(wid == 0){
		const wtz_tb_score sc = { M, X, O, E, init_score };
		const bool cons = wtz_shift_traceback<C, NL, false>(x, zchunk, zb, zrow, tb, cigars, &sc);
		if(!cons && lane == 0) sh->flag = 1;
		x = wtz_bcast_aln(x);
	}
	__syncthreads();
	if(sh->flag){ *ok = false; *consistent = false; }
	return x;
}

/* one K-sw3 job on the calling WORKGROUP of 256 threads; false (the same in every thread) = outside the envelope, or its counts did not follow from the score:
 * the job stays open for the other forms.  stb: TW 64-bit words, shm: the exchange block (both LDS of the workgroup). */
template<int TW>
WTZ_D bool wtz_extjob_run_frmw(wtz_extjob_t *job, const wtz_params_t *Pm, wtz_pool_t *pool, wtz_pool_t *tpool, uint64_t *stb, wtz_frmw_shared_t *shm){
	if(!job->valid || job->done) return true;
	const int tid = (int)threadIdx.x;
	if(job->qlen <= 0 || job->tlen <= 0) return false;
	const int32_t init_score = job->init_score < 0 ? 0 : job->init_score;
	int32_t W = job->W, ql, tl, n_col;
	wtz_ext_geometry(job->qlen, job->tlen, init_score, W, Pm->M, Pm->O, Pm->O, Pm->E, Pm->T, ql, tl, n_col);
	const int32_t Cw = (n_col + 255) / 256;
	if(Cw > 8 || (tl + 63) / 32 + 1 > TW || (ql + 63) / 64 > WTZ_TRACE_MAXCHUNK) return false;
	{
		const long long aE = Pm->E < 0 ? -(long long)Pm->E : (long long)Pm->E, aX = Pm->X < 0 ? -(long long)Pm->X : (long long)Pm->X, aO = Pm->O < 0 ? -(long long)Pm->O : (long long)Pm->O;
		const long long span = (long long)ql + tl + 4;
		if((long long)init_score + (long long)(Pm->M > 0 ? Pm->M : -Pm->M) * (ql < tl ? ql : tl) + span * aE + 10000 + aX + aO + 16 >= (1 << 20)) return false;
		if(aE > 255 || Pm->M == Pm->X) return false;
	}
	wtz_cigar_t cg; cg.a = NULL; cg.n = cg.cap = 0; cg.pool = pool; cg.bad = 0;
	if(tid == 0) cg.init(pool, (uint32_t)ql / 2u + 16u);
	unsigned long long cells = 0; bool ok = true, consistent = true;
	wtz_aln_t x;
#define WTZ_EXTFRMW_CASE(CM) x = wtz_extend_shift_frmw<CM, 4>(job->qlen, job->q, job->tlen, job->t, job->init_score, ql, tl, W, Pm->M, Pm->X, Pm->O, Pm->E, Pm->T, stb, shm, tpool, cg, &cells, &ok, &consistent)
	if(Cw <= 2) WTZ_EXTFRMW_CASE(2);
	else if(Cw <= 4) WTZ_EXTFRMW_CASE(4);
	else if(Cw <= 6) WTZ_EXTFRMW_CASE(6);
	else WTZ_EXTFRMW_CASE(8);
#undef WTZ_EXTFRMW_CASE
	if(!consistent) return false;
	if(tid == 0){ job->x = x; job->cigar = cg.a; job->cigar_len = cg.n; job->bad = (!ok || cg.bad); job->cells = cells; job->done = 6; }
	return true;
}

template<int TW>
__global__ void __launch_bounds__(256) wtz_kernel_extjobs_frmw(wtz_extjob_t *jobs, const uint32_t *order, uint32_t n, const wtz_params_t *Pm, wtz_pool_t *pool, wtz_pool_t *tpool){
	__shared__ uint64_t stb[TW]; __shared__ wtz_frmw_shared_t shm;
	const uint32_t b = blockIdx.x;
	if(b >= n) return;
	(void)wtz_extjob_run_frmw<TW>(&jobs[order ? order[b] : b], Pm, pool, tpool, stb, &shm);
}

#endif /* __HIPCC__ */
#endif
