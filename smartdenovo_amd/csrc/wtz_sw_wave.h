/*
 * wtz_sw_wave.h — the extension DPs with one 64-lane wavefront per problem:
 *     MODE 0  K-sw3  kswx_extend_align_shift_core  kswx.h:101-232   (band follows the row arg-max)
 *     MODE 1  K-sw1  kswx_extend_align_core        kswx.h:234-335   (fixed band around the diagonal)
 *
 * Why row-wise and not anti-diagonal: in K-sw3 the band centre of row i+1 depends on the arg-max of row i
 * (kswx.h:186-199), so rows are inherently sequential; inside a row, however, the reference's recurrence opens
 * gaps from m (the diagonal value), not from H:
 *        m_j = H(i-1,j-1) + S          E'_j = max(E_j + e, m_j + I + e)        F_{j+1} = max(F_j + e, m_j + D + e)
 * so F along the row is a max-plus prefix scan of values that depend on the previous row only, exact in int32.
 * A row is therefore: (1) every lane computes m for its C consecutive columns and a local scan aggregate,
 * (2) one 6-step cross-lane max-scan over the 64 lane aggregates gives each lane its carry-in, (3) every lane
 * finishes H / E' / F / trace bits for its columns, (4) a 6-step (value, column) reduction yields the row maximum
 * and its arg-max (FIRST for K-sw3, LAST for K-sw1: kswx.h:172 vs 288-289), which moves the K-sw3 band.
 *
 * Data layout (per wave, LDS): Hs[P], Es[P] int32 rings indexed by absolute column & (P-1) (P >= band+2); lanes own
 * column blocks of odd width C so the stride-C ds_read/ds_write are bank-conflict free; the target segment is staged
 * once as 2-bit codes in logical order (strand / complement resolved) so that a lane fetches its <= 31 bases with two
 * ds_read_b64.  Trace bytes go to HBM in a lane-transposed layout (row, kk/4, lane, kk%4): each of the ceil(C/4)
 * stores per row writes 256 contiguous bytes; rows are carved from the pool 64 at a time (most extensions stop long
 * before ql rows) and a task reuses its chunks across the many small K-sw1 problems of a window.
 * Semantics reproduced exactly: -10000 sentinels outside the band, H(i,-1)/H(-1,j) boundary values, arg-max tie
 * rules, gmax/max end rule with T, early exit when a row maximum is <= 0, the per-row band start of K-sw3.
 */
#ifndef WTZ_SW_WAVE_H
#define WTZ_SW_WAVE_H

#include "wtz_sw.h"

typedef struct {
	wtz_seq_packed q, t;        /* logical views: index 0 is the base next to the seed, walking outwards */
	int32_t qlen, tlen, init_score, W;      /* W as passed to kswx_extend_align_shift_core (negative = exact band) */
	uint32_t item;              /* owner (stitch item) */
	uint32_t valid;             /* 0: nothing to do */
	uint32_t done;              /* set by the register-DP kernel; the general kernel then skips the job */
	/* results */
	wtz_aln_t x; uint32_t *cigar; uint32_t cigar_len; int32_t bad; unsigned long long cells;
} wtz_extjob_t;

#ifndef WTZ_OCC_EXTREG
#define WTZ_OCC_EXTREG 1
#endif
#ifndef WTZ_WINALIGN_LDS_BYTES
#define WTZ_WINALIGN_LDS_BYTES 12288     /* LDS slice of a window-alignment wave (measured best with 3 waves/SIMD) */
#endif
#ifndef WTZ_WINALIGN_QW_BYTES
#define WTZ_WINALIGN_QW_BYTES 512       /* behind the slice: the query of the current K-sw1 problem as 2-bit words, for its traceback */
#endif

#ifdef __HIPCC__

#define WTZ_TRACE_MAXCHUNK 1024          /* 64-row chunks: up to 65536 rows per problem */

/* reusable trace storage of one task (pointers live in the pool so that lane 0 can chase them during traceback) */
typedef struct { uint8_t **chunk; int32_t *zb; uint32_t n_chunk, zrow, cap_rows; } wtz_trace_t;

/* LDS view handed to the wave DP: rings of PM+1 ints (PM = size-1 mask), target buffer of tw 64-bit words */
typedef struct { int32_t *Hs, *Es; uint64_t *tb; int32_t PM; int32_t tw; uint32_t *qw; } wtz_wave_lds_t;      /* qw: 512 B for the query words of the K-sw1 traceback (only wtz_fixed_problem_wave reads it) */

/* ---- wavefront-wide max scan / arg-max reduction on the DPP data path (no LDS round trip) ----
 * gfx9-family DPP controls: row_shr:n = 0x110+n (inside a row of 16 lanes), row_bcast:15 = 0x142 (lane 15 of each row to
 * the next row), row_bcast:31 = 0x143 (lane 31 to rows 2-3), wave_shr:1 = 0x138.  Lanes without a source keep `old`. */
template<int CTRL, int ROWMASK>
WTZ_D int32_t wtz_dpp_mov(int32_t old, int32_t src){ return __builtin_amdgcn_update_dpp(old, src, CTRL, ROWMASK, 0xF, false); }

/* neighbour lane's value: wave_shl:1 = from lane+1, wave_shr:1 = from lane-1; the edge lane keeps `old` */
WTZ_D int32_t wtz_dpp_wave_shl1(int32_t old, int32_t src){ return __builtin_amdgcn_update_dpp(old, src, 0x130, 0xF, 0xF, false); }
WTZ_D int32_t wtz_dpp_wave_shr1(int32_t old, int32_t src){ return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xF, 0xF, false); }

/* exclusive prefix maximum over the lanes; lane 0 gets `ident`.  The inclusive steps give a lane without a source INT_MIN, the
 * identity of max (every caller's values are >= ident > INT_MIN, so the result is the same as with `ident`): the compiler's DPP
 * combiner then folds each step into ONE v_max_i32_dpp instead of v_mov (ident) + v_mov_dpp + v_max - 12 VALU ops less per DP
 * row of every register DP */
WTZ_D int32_t wtz_wave_max_scan_excl(int32_t v, int32_t ident){
	const int32_t mn = (int32_t)0x80000000;
	int32_t x = v, t;
	t = wtz_dpp_mov<0x111, 0xF>(mn, x); x = x > t ? x : t;
	t = wtz_dpp_mov<0x112, 0xF>(mn, x); x = x > t ? x : t;
	t = wtz_dpp_mov<0x114, 0xF>(mn, x); x = x > t ? x : t;
	t = wtz_dpp_mov<0x118, 0xF>(mn, x); x = x > t ? x : t;
	t = wtz_dpp_mov<0x142, 0xA>(mn, x); x = x > t ? x : t;
	t = wtz_dpp_mov<0x143, 0xC>(mn, x); x = x > t ? x : t;
	return wtz_dpp_mov<0x138, 0xF>(ident, x);              /* inclusive -> exclusive: shift the whole wave by one lane */
}

/* all-lanes maximum of an int32, result uniform */
WTZ_D int32_t wtz_wave_max_i32(int32_t v){
	const int32_t ident = (int32_t)0x80000000;
	int32_t x = v, t;
	t = wtz_dpp_mov<0x111, 0xF>(ident, x); x = x > t ? x : t;
	t = wtz_dpp_mov<0x112, 0xF>(ident, x); x = x > t ? x : t;
	t = wtz_dpp_mov<0x114, 0xF>(ident, x); x = x > t ? x : t;
	t = wtz_dpp_mov<0x118, 0xF>(ident, x); x = x > t ? x : t;
	t = wtz_dpp_mov<0x142, 0xA>(ident, x); x = x > t ? x : t;
	t = wtz_dpp_mov<0x143, 0xC>(ident, x); x = x > t ? x : t;
	return __builtin_amdgcn_readlane(x, 63);
}

/* all-lanes maximum of a signed 64-bit key (hi:int32 value, lo:uint32 tie-break), result in every lane */
WTZ_D void wtz_wave_max_key(int32_t &hi, uint32_t &lo){
	const int32_t ih = (int32_t)0x80000000; const int32_t il = 0;
#define WTZ_KEYSTEP(CTRL, RM) { const int32_t th = wtz_dpp_mov<CTRL, RM>(ih, hi); const uint32_t tl = (uint32_t)wtz_dpp_mov<CTRL, RM>(il, (int32_t)lo); \
	if(th > hi || (th == hi && tl > lo)){ hi = th; lo = tl; } }
	WTZ_KEYSTEP(0x111, 0xF) WTZ_KEYSTEP(0x112, 0xF) WTZ_KEYSTEP(0x114, 0xF) WTZ_KEYSTEP(0x118, 0xF) WTZ_KEYSTEP(0x142, 0xA) WTZ_KEYSTEP(0x143, 0xC)
#undef WTZ_KEYSTEP
	hi = __builtin_amdgcn_readlane(hi, 63); lo = (uint32_t)__builtin_amdgcn_readlane((int32_t)lo, 63);
}

WTZ_D bool wtz_wave_fits(const wtz_wave_lds_t &L, int32_t n_col, int32_t tl, int32_t ql){
	return n_col + 2 <= L.PM + 1 && n_col <= 64 * 31 && (tl + 63) / 32 + 1 <= L.tw && (ql + 63) / 64 <= WTZ_TRACE_MAXCHUNK;
}

WTZ_D wtz_aln_t wtz_bcast_aln(wtz_aln_t x){
	wtz_aln_t r;
	r.score = __shfl(x.score, 0, 64); r.tb = __shfl(x.tb, 0, 64); r.te = __shfl(x.te, 0, 64); r.qb = __shfl(x.qb, 0, 64); r.qe = __shfl(x.qe, 0, 64);
	r.aln = __shfl(x.aln, 0, 64); r.mat = __shfl(x.mat, 0, 64); r.mis = __shfl(x.mis, 0, 64); r.ins = __shfl(x.ins, 0, 64); r.del = __shfl(x.del, 0, 64);
	return r;
}

/* make sure the trace has chunks for `zrow`-byte rows; lane 0 allocates, everybody learns the table address.
 * returns false on pool exhaustion (uniform) */
WTZ_D bool wtz_trace_prepare(wtz_trace_t &tr, wtz_pool_t *pool, uint32_t zrow, int32_t ql, bool need_zb){
	const int lane = (int)(threadIdx.x & 63);
	if(tr.chunk == NULL || tr.zrow != zrow){
		unsigned long long a = 0;
		if(lane == 0) a = (unsigned long long)(uintptr_t)wtz_pool_alloc(pool, (size_t)WTZ_TRACE_MAXCHUNK * 8);
		a = __shfl(a, 0, 64);
		tr.chunk = (uint8_t**)(uintptr_t)a; tr.n_chunk = 0; tr.zrow = zrow;
		if(tr.chunk == NULL) return false;
	}
	if(need_zb && (tr.zb == NULL || tr.cap_rows < (uint32_t)ql + 2)){
		unsigned long long a = 0;
		if(lane == 0) a = (unsigned long long)(uintptr_t)wtz_pool_alloc(pool, (size_t)(ql + 2) * 4);
		a = __shfl(a, 0, 64);
		tr.zb = (int32_t*)(uintptr_t)a; tr.cap_rows = (uint32_t)ql + 2;
		if(tr.zb == NULL) return false;
	}
	return true;
}

/*
 * The DP.  All 64 lanes call it with identical arguments; the result x is identical on all lanes; the CIGAR is
 * pushed into `cigars` on lane 0 only (cleared first).  *ok is false when the pool ran dry.
 */
template<int MODE, typename SQ, typename ST>
WTZ_D wtz_aln_t wtz_extend_wave(int32_t qlen, const SQ &query, int32_t tlen, const ST &target, int32_t init_score, int32_t W,
		int32_t M, int32_t X, int32_t I, int32_t D, int32_t E, int32_t T, const wtz_wave_lds_t &L, wtz_trace_t &tr, wtz_pool_t *pool,
		wtz_cigar_t &cigars, unsigned long long *cells, bool *ok){
	const int lane = (int)(threadIdx.x & 63);
	const int32_t PM = L.PM;
	wtz_aln_t x; memset(&x, 0, sizeof x);
	*ok = true;
	if(lane == 0) cigars.n = 0;
	if(init_score < 0) init_score = 0;
	if(qlen <= 0 || tlen <= 0){ x.score = init_score; return x; }
	int32_t ql, tl, n_col;
	wtz_ext_geometry(qlen, tlen, init_score, W, M, I, D, E, T, ql, tl, n_col);
	const int32_t C0 = (n_col + 63) / 64, C = (C0 | 1);          /* odd block width: conflict-free LDS stride */
	const int32_t C4 = (C + 3) / 4;                               /* trace dwords per lane per row */
	const uint32_t zrow = (uint32_t)C4 * 256u;
	if(!wtz_trace_prepare(tr, pool, zrow, ql, MODE == 0)){ *ok = false; return x; }
	uint8_t **zchunk = tr.chunk; int32_t *zb = tr.zb;
	uint8_t *z = NULL;
	/* stage the target segment [0, tl) as 2-bit codes */
	{
		const int32_t nw = (tl + 31) / 32 + 1;
		for(int32_t w = lane; w < nw; w += 64){
			uint64_t v = 0; const int32_t b0 = w * 32;
			for(int32_t k = 0; k < 32 && b0 + k < tl; k++) v |= ((uint64_t)target.at(b0 + k)) << (2 * k);
			L.tb[w] = v;
		}
	}
	int32_t mx = init_score, mi = -1, mj = -1, gmax = 0, gi = -1, gj = -1;
	int32_t jbp = 0, jep = 0;                 /* band of the previous row: Hs/Es valid on [jbp, jep) */
	int32_t c = 0, i;
	unsigned long long ncell = 0;
	const int32_t CE = C * E;
	for(i = 0; i < ql; i++){
		if((i & 63) == 0){
			const uint32_t ci = (uint32_t)i >> 6;
			unsigned long long za = 0;
			if(ci < tr.n_chunk){ z = zchunk[ci]; }
			else {
				if(lane == 0){ uint8_t *p = (uint8_t*)wtz_pool_alloc(pool, (size_t)zrow * 64); zchunk[ci] = p; za = (unsigned long long)(uintptr_t)p; }
				za = __shfl(za, 0, 64);
				z = (uint8_t*)(uintptr_t)za;
				if(z == NULL){ *ok = false; break; }
				tr.n_chunk = ci + 1;
			}
		}
		int32_t jb, je;
		if(MODE == 0){
			jb = 0; je = tl;
			if(jb < c - W) jb = c - W;
			if(je > c + W + 1) je = c + W + 1;
			if(je > tl) je = tl;
		} else {
			jb = i - W; if(jb < 0) jb = 0;
			je = i + W + 1; if(je > tl) je = tl;
		}
		const uint32_t qbase = query.at(i);
		const int32_t j0 = jb + lane * C;                /* first column of this lane */
		uint64_t tbits;                                   /* the lane's target bases: columns j0 .. j0+C-1 (<= 31) */
		{
			const int32_t jj = j0 < tl ? j0 : (tl > 0 ? tl - 1 : 0);
			const int32_t w = jj >> 5, sh = (jj & 31) * 2;
			const uint64_t w0 = L.tb[w], w1 = L.tb[w + 1];
			tbits = sh ? ((w0 >> sh) | (w1 << (64 - sh))) : w0;
		}
		/* ---- pass 1: m_j into Hs[j], local F aggregate ---- */
		int32_t agg = -0x3FFFFFFF;          /* max_k ( m_k + D + E + (C-1-k)*E ) */
		{
			int32_t saved = 0;
			for(int32_t k = 0; k < C; k++){
				const int32_t j = j0 + k;
				if(j < je){
					int32_t pred;
					if(k == 0){
						if(i == 0) pred = (j == 0) ? init_score : init_score + D + E * j;          /* rh[] initialisation, kswx.h:143-144 */
						else if(j - 1 >= jbp && j - 1 < jep) pred = L.Hs[(j - 1) & PM];
						else pred = (j == 0) ? init_score + I + E * i : -10000;
					} else {
						if(i == 0) pred = init_score + D + E * j;
						else pred = saved;
					}
					saved = (i > 0 && j >= jbp && j < jep) ? L.Hs[j & PM] : -10000;       /* H(i-1, j) for the next column, before it is overwritten */
					const uint32_t tbase = (uint32_t)(tbits >> (2 * k)) & 3u;
					const int32_t m = pred + ((qbase == tbase) ? M : X);
					L.Hs[j & PM] = m;
					const int32_t cand = m + D + E + (C - 1 - k) * E;
					agg = agg > cand ? agg : cand;
				}
			}
		}
		/* ---- cross-lane exclusive max-plus scan: carry-in F for the lane's first column ---- */
		int32_t f_in;
		{
			const int32_t g = agg - lane * CE;
			const int32_t pm = wtz_wave_max_scan_excl(g, -0x3FFFFFFF);
			const int32_t from_prev = (lane == 0) ? -0x3FFFFFFF : pm + (lane - 1) * CE;
			const int32_t from_init = -10000 + lane * CE;                     /* F at column jb is -10000 (kswx.h:157) */
			f_in = from_prev > from_init ? from_prev : from_init;
		}
		/* ---- pass 2: H, E', F, trace ---- */
		int32_t best = -0x7FFFFFFF, bestj = (MODE == 0) ? 0x7FFFFFFF : -1, h_last = 0;
		{
			int32_t f = f_in; uint32_t zword = 0;
			uint32_t *zr = (uint32_t*)(z + (size_t)(i & 63) * zrow) + lane;
			for(int32_t k = 0; k < C; k++){
				const int32_t j = j0 + k;
				if(j < je){
					const int32_t m = L.Hs[j & PM];
					int32_t e = (j >= jbp && j < jep) ? L.Es[j & PM] : -10000;
					uint32_t d; int32_t h;
					if(m >= e){ d = 0; h = m; } else { d = 1; h = e; }
					if(h < f){ d = 2; h = f; }
					if(MODE == 0){ if(h > best){ best = h; bestj = j; } }
					else         { if(h >= best){ best = h; bestj = j; } }
					h_last = h;
					int32_t t = m + I + E; e = e + E; if(e > t) d |= 1u << 2; else e = t;
					t = m + D + E; f = f + E; if(f > t) d |= 2u << 4; else f = t;
					L.Hs[j & PM] = h; L.Es[j & PM] = e;
					zword |= d << (8 * (k & 3));
				}
				if((k & 3) == 3 || k == C - 1){ zr[(size_t)(k >> 2) * 64] = zword; zword = 0; }
			}
		}
		ncell += (unsigned long long)(je - jb);
		/* ---- row maximum and its arg-max ---- */
		if(MODE == 0){ uint32_t lo = 0x7FFFFFFFu - (uint32_t)bestj; wtz_wave_max_key(best, lo); bestj = (int32_t)(0x7FFFFFFFu - lo); }   /* smallest column */
		else         { uint32_t lo = (uint32_t)(bestj + 1); wtz_wave_max_key(best, lo); bestj = (int32_t)lo - 1; }                          /* largest column */
		int32_t imax = 0, mj2 = -1;
		if(MODE == 0){ if(best > 0){ imax = best; mj2 = bestj; } }            /* first j with the maximum, only if > 0 (kswx.h:172) */
		else         { if(best >= 0){ imax = best; mj2 = bestj; } }           /* last j with h >= running max >= 0 (kswx.h:288-289) */
		const int32_t lastlane = (je - 1 - jb) / C;
		const int32_t h1 = __builtin_amdgcn_readlane(h_last, __builtin_amdgcn_readfirstlane(lastlane));                      /* H(i, je-1) */
		if(MODE == 0 && lane == 0) zb[i] = jb;
		if(je == tlen && gmax < h1){ gmax = h1; gi = i; gj = je - 1; }
		if(i + 1 == qlen && gmax < imax){ gmax = imax; gi = i; gj = mj2; }
		jbp = jb; jep = je;
		if(imax > mx){ mx = imax; mi = i; mj = mj2; }
		else if(imax <= 0) break;
		if(MODE == 0){ c++; if(c < mj2) c++; else if(c > mj2) c--; }
	}
	if(cells && lane == 0) *cells += ncell;
	if(!*ok) return x;
	if(gmax > 0 && gmax >= mx + T){ x.score = gmax; x.qe = gi; x.te = gj; }
	else { x.score = mx; x.qe = mi; x.te = mj; }
	/* ---- traceback by lane 0 (reads trace bytes stored by the other lanes of this wave) ---- */
	__threadfence_block();     /* the trace was written by lanes of THIS wave: ordering inside the wave is enough (an agent-scope fence would write back the whole L2) */
	if(lane == 0){
		int32_t i_ = x.qe, j_ = x.te; uint32_t d_ = 0;
		while(i_ >= 0 && j_ >= 0){
			const int32_t rowb = (MODE == 0) ? zb[i_] : (i_ > W ? i_ - W : 0);
			const int32_t col = j_ - rowb;
			const int32_t ln = col / C, kk = col - ln * C;
			const uint8_t zv = zchunk[i_ >> 6][(size_t)(i_ & 63) * zrow + (size_t)(kk >> 2) * 256 + (size_t)ln * 4 + (kk & 3)];
			d_ = (zv >> (d_ << 1)) & 0x03;
			if(d_ == 0){ if(query.at(i_) == target.at(j_)) x.mat++; else x.mis++; i_--; j_--; }
			else if(d_ == 1){ i_--; x.ins++; }
			else { j_--; x.del++; }
			wtz_cigar_push(cigars, d_, 1);
		}
		if(i_ >= 0){ x.ins += i_ + 1; wtz_cigar_push(cigars, 1, (uint32_t)(i_ + 1)); }
		if(j_ >= 0){ x.del += j_ + 1; wtz_cigar_push(cigars, 2, (uint32_t)(j_ + 1)); }
		wtz_cigar_reverse(cigars.a, cigars.n);
		x.aln = x.mat + x.mis + x.ins + x.del; x.qe++; x.te++;
	}
	return wtz_bcast_aln(x);
}

/*
 * K-sw3, register-tiled rows.  Same algorithm and results as wtz_extend_wave<0>, but a lane keeps its block of the
 * previous row (H and E of <= CMAX columns) in VGPRs: per row it issues all its LDS reads back to back (one
 * latency), computes m / F-scan / H / E' / trace entirely in registers, and issues all LDS writes back to back.
 * The LDS rings are only the hand-over medium between rows (the band moves, so column ownership changes).
 * This cuts the row latency - which is what bounds the longest extension of a batch - by the C dependent
 * ds_read -> ds_write round trips of the loop form.
 */
template<int CMAX, typename SQ, typename ST>
WTZ_D wtz_aln_t wtz_extend_shift_wave_rt(int32_t qlen, const SQ &query, int32_t tlen, const ST &target, int32_t init_score, int32_t W,
		int32_t M, int32_t X, int32_t I, int32_t D, int32_t E, int32_t T, const wtz_wave_lds_t &L, wtz_trace_t &tr, wtz_pool_t *pool,
		wtz_cigar_t &cigars, unsigned long long *cells, bool *ok){
	const int lane = (int)(threadIdx.x & 63);
	const int32_t PM = L.PM;
	wtz_aln_t x; memset(&x, 0, sizeof x);
	*ok = true;
	if(lane == 0) cigars.n = 0;
	if(init_score < 0) init_score = 0;
	if(qlen <= 0 || tlen <= 0){ x.score = init_score; return x; }
	int32_t ql, tl, n_col;
	wtz_ext_geometry(qlen, tlen, init_score, W, M, I, D, E, T, ql, tl, n_col);
	const int32_t C0 = (n_col + 63) / 64, C = (C0 | 1);
	const int32_t C4 = (C + 3) / 4;
	const uint32_t zrow = (uint32_t)C4 * 256u;
	if(!wtz_trace_prepare(tr, pool, zrow, ql, true)){ *ok = false; return x; }
	uint8_t **zchunk = tr.chunk; int32_t *zb = tr.zb;
	uint8_t *z = NULL;
	{
		const int32_t nw = (tl + 31) / 32 + 1;
		for(int32_t w = lane; w < nw; w += 64){
			uint64_t v = 0; const int32_t b0 = w * 32;
			for(int32_t k = 0; k < 32 && b0 + k < tl; k++) v |= ((uint64_t)target.at(b0 + k)) << (2 * k);
			L.tb[w] = v;
		}
	}
	int32_t mx = init_score, mi = -1, mj = -1, gmax = 0, gi = -1, gj = -1;
	int32_t jbp = 0, jep = 0, c = 0, i;
	unsigned long long ncell = 0;
	const int32_t CE = C * E;
	for(i = 0; i < ql; i++){
		if((i & 63) == 0){
			const uint32_t ci = (uint32_t)i >> 6;
			unsigned long long za = 0;
			if(ci < tr.n_chunk){ z = zchunk[ci]; }
			else {
				if(lane == 0){ uint8_t *p = (uint8_t*)wtz_pool_alloc(pool, (size_t)zrow * 64); zchunk[ci] = p; za = (unsigned long long)(uintptr_t)p; }
				za = __shfl(za, 0, 64);
				z = (uint8_t*)(uintptr_t)za;
				if(z == NULL){ *ok = false; break; }
				tr.n_chunk = ci + 1;
			}
		}
		int32_t jb = 0, je = tl;
		if(jb < c - W) jb = c - W;
		if(je > c + W + 1) je = c + W + 1;
		if(je > tl) je = tl;
		const uint32_t qbase = query.at(i);
		const int32_t j0 = jb + lane * C;
		uint64_t tbits;
		{
			const int32_t jj = j0 < tl ? j0 : (tl > 0 ? tl - 1 : 0);
			const int32_t w = jj >> 5, sh = (jj & 31) * 2;
			const uint64_t w0 = L.tb[w], w1 = L.tb[w + 1];
			tbits = sh ? ((w0 >> sh) | (w1 << (64 - sh))) : w0;
		}
		/* ---- load phase: previous-row H and E of the lane's columns, plus H(i-1, j0-1) ---- */
		int32_t hv[CMAX], ev[CMAX];
		int32_t pred0;
		if(i == 0) pred0 = (j0 == 0) ? init_score : init_score + D + E * j0;
		else if(j0 - 1 >= jbp && j0 - 1 < jep) pred0 = L.Hs[(j0 - 1) & PM];
		else pred0 = (j0 == 0) ? init_score + I + E * i : -10000;
		#pragma unroll
		for(int k = 0; k < CMAX; k++){
			const int32_t j = j0 + k;
			const bool live = (k < C) && (i > 0) && (j >= jbp) && (j < jep);
			hv[k] = live ? L.Hs[j & PM] : -10000;
			ev[k] = live ? L.Es[j & PM] : -10000;
		}
		/* ---- m in place (descending k: m_k needs the OLD H of column k-1) and the local F aggregate ---- */
		int32_t agg = -0x3FFFFFFF;
		#pragma unroll
		for(int k = CMAX - 1; k >= 0; k--){
			const int32_t j = j0 + k;
			int32_t pred;
			if(k == 0) pred = pred0;
			else pred = (i == 0) ? init_score + D + E * j : hv[k - 1];
			const uint32_t tbase = (uint32_t)(tbits >> (2 * k)) & 3u;
			const int32_t m = pred + ((qbase == tbase) ? M : X);
			hv[k] = m;
			if(k < C && j < je){ const int32_t cand = m + D + E + (C - 1 - k) * E; agg = agg > cand ? agg : cand; }
		}
		int32_t f;
		{
			const int32_t g = agg - lane * CE;
			const int32_t pm = wtz_wave_max_scan_excl(g, -0x3FFFFFFF);
			const int32_t from_prev = (lane == 0) ? -0x3FFFFFFF : pm + (lane - 1) * CE;
			const int32_t from_init = -10000 + lane * CE;
			f = from_prev > from_init ? from_prev : from_init;
		}
		/* ---- H, E', F, trace in registers ---- */
		int32_t best = -0x7FFFFFFF, bestj = 0x7FFFFFFF, h_last = 0;
		uint32_t zw[(CMAX + 3) / 4];
		#pragma unroll
		for(int q4 = 0; q4 < (CMAX + 3) / 4; q4++) zw[q4] = 0;
		#pragma unroll
		for(int k = 0; k < CMAX; k++){
			const int32_t j = j0 + k;
			if(k < C && j < je){
				const int32_t m = hv[k];
				int32_t e = ev[k];
				uint32_t d; int32_t h;
				if(m >= e){ d = 0; h = m; } else { d = 1; h = e; }
				if(h < f){ d = 2; h = f; }
				if(h > best){ best = h; bestj = j; }
				h_last = h;
				int32_t t = m + I + E; e = e + E; if(e > t) d |= 1u << 2; else e = t;
				t = m + D + E; f = f + E; if(f > t) d |= 2u << 4; else f = t;
				hv[k] = h; ev[k] = e;
				if(qbase == ((uint32_t)(tbits >> (2 * k)) & 3u)) d |= 0x80u;       /* bit 7: bases equal (saves two sequence loads per traceback step) */
				zw[k >> 2] |= d << (8 * (k & 3));
			}
		}
		/* ---- store phase ---- */
		#pragma unroll
		for(int k = 0; k < CMAX; k++){
			const int32_t j = j0 + k;
			if(k < C && j < je){ L.Hs[j & PM] = hv[k]; L.Es[j & PM] = ev[k]; }
		}
		{
			uint32_t *zr = (uint32_t*)(z + (size_t)(i & 63) * zrow) + lane;
			#pragma unroll
			for(int q4 = 0; q4 < (CMAX + 3) / 4; q4++) if(q4 < C4) zr[(size_t)q4 * 64] = zw[q4];
		}
		ncell += (unsigned long long)(je - jb);
		{ uint32_t lo = 0x7FFFFFFFu - (uint32_t)bestj; wtz_wave_max_key(best, lo); bestj = (int32_t)(0x7FFFFFFFu - lo); }     /* max value, then smallest column */
		int32_t imax = 0, mj2 = -1;
		if(best > 0){ imax = best; mj2 = bestj; }
		const int32_t lastlane = (je - 1 - jb) / C;
		const int32_t h1 = __builtin_amdgcn_readlane(h_last, __builtin_amdgcn_readfirstlane(lastlane));
		if(lane == 0) zb[i] = jb;
		if(je == tlen && gmax < h1){ gmax = h1; gi = i; gj = je - 1; }
		if(i + 1 == qlen && gmax < imax){ gmax = imax; gi = i; gj = mj2; }
		jbp = jb; jep = je;
		if(imax > mx){ mx = imax; mi = i; mj = mj2; }
		else if(imax <= 0) break;
		c++; if(c < mj2) c++; else if(c > mj2) c--;
	}
	if(cells && lane == 0) *cells += ncell;
	if(!*ok) return x;
	if(gmax > 0 && gmax >= mx + T){ x.score = gmax; x.qe = gi; x.te = gj; }
	else { x.score = mx; x.qe = mi; x.te = mj; }
	__threadfence_block();     /* the trace was written by lanes of THIS wave: ordering inside the wave is enough (an agent-scope fence would write back the whole L2) */
	if(lane == 0){
		/* traceback: one dependent trace-byte load per step; the band start of the previous row is fetched alongside, the chunk
		 * base only every 64 rows, match/mismatch comes from bit 7 of the byte, the CIGAR run is kept in registers */
		int32_t i_ = x.qe, j_ = x.te; uint32_t d_ = 0;
		int32_t chunk_i = -1; const uint8_t *cbase = NULL;
		int32_t zbi = (i_ >= 0) ? zb[i_] : 0;
		uint32_t run_op = 0xFFu, run_len = 0;
		while(i_ >= 0 && j_ >= 0){
			if((i_ >> 6) != chunk_i){ chunk_i = i_ >> 6; cbase = zchunk[chunk_i]; }
			const int32_t col = j_ - zbi;
			const int32_t ln = col / C, kk = col - ln * C;
			const uint8_t zv = cbase[(size_t)(i_ & 63) * zrow + (size_t)(kk >> 2) * 256 + (size_t)ln * 4 + (kk & 3)];
			const int32_t zb_prev = (i_ > 0) ? zb[i_ - 1] : 0;
			d_ = (zv >> (d_ << 1)) & 0x03;
			if(d_ == 0){ if(zv & 0x80u) x.mat++; else x.mis++; i_--; j_--; zbi = zb_prev; }
			else if(d_ == 1){ i_--; x.ins++; zbi = zb_prev; }
			else { j_--; x.del++; }
			if(d_ == run_op) run_len++;
			else { if(run_len) wtz_cigar_push(cigars, run_op, run_len); run_op = d_; run_len = 1; }
		}
		if(run_len) wtz_cigar_push(cigars, run_op, run_len);
		if(i_ >= 0){ x.ins += i_ + 1; wtz_cigar_push(cigars, 1, (uint32_t)(i_ + 1)); }
		if(j_ >= 0){ x.del += j_ + 1; wtz_cigar_push(cigars, 2, (uint32_t)(j_ + 1)); }
		wtz_cigar_reverse(cigars.a, cigars.n);
		x.aln = x.mat + x.mis + x.ins + x.del; x.qe++; x.te++;
	}
	return wtz_bcast_aln(x);
}

/* lane-0 CIGAR writer: the open run stays in a register, so appending never reads memory back (kswx_push_cigar merges) */
typedef struct { wtz_cigar_t *v; uint32_t tail; } wtz_cigw_t;
WTZ_D void wtz_cigw_push(wtz_cigw_t &w, uint32_t op, uint32_t len){
	if(len == 0) return;
	if(w.tail && (w.tail & 0xFu) == op) w.tail += len << 4;
	else { if(w.tail) w.v->push(w.tail); w.tail = (len << 4) | op; }
}
WTZ_D void wtz_cigw_finish(wtz_cigw_t &w){ if(w.tail){ w.v->push(w.tail); w.tail = 0; } }

/* ---- traceback of the K-sw3 register forms (one wavefront; NL = lanes of a trace row: 64, or 256 for the four-wave form).
 * The trace lives in the pool in the lane-transposed layout (row, k/4, lane, k%4).  Lane 0 walks, but never against HBM latency:
 * for the 64 rows below the current cell the wave copies the dwords of the NLW lanes around the current band-relative column
 * into LDS - one row per step, lane x takes dword (x / C4, x % C4) of the row, so a step is a few contiguous pieces and the loads
 * of eight rows are in flight together (the first version fetched byte by byte, each byte waiting for its own round trip: a 64-row
 * block cost more than the 64 DP rows it traces) - decoded four cells at a time into the walker's bytes, 128 bytes per row in
 * band-relative column order.  The walker tracks the band-relative column itself (a row's band start moves by 0..2 against the
 * row above: 64 deltas in LDS), leaves the block after 64 rows or through either edge of the window, and the next block is
 * staged around the cell it stands on.  Runs go through the register-tail writer and are reversed once at the end.
 * LDS: 8192 + 64 bytes at `tb` (the target words are dead by now). ---- */
/* EQ = false (the frame form, wtz_sw_frame.h): the trace byte holds the four decisions only (bit 3 m<e, bit 2 max(m,e)<f, bit 1 E extended, bit 0 F extended); the walk
 * counts diagonal steps and gap runs, and matches / mismatches follow from the score: every gap run of the path was opened once (a run ends in the diagonal step
 * of the cell it was opened from - kswx.h:175-183 opens from m, not from H - so two runs never touch), hence
 *     score - init = M*mat + X*mis + runs*O + E*(ins + del),   mat + mis = diagonal steps.
 * `sc` = {M, X, O, E, init_score (clamped)}; returns false when the identity does not come out in integers (never observed; the caller then leaves the job to the general kernel). */
struct wtz_tb_score { int32_t M, X, O, E, init; };
template<int C, int NL, bool EQ = true>
WTZ_D bool wtz_shift_traceback(wtz_aln_t &x, uint8_t **zchunk, const int32_t *zb, uint32_t zrow, uint64_t *tb, wtz_cigar_t &cigars, const wtz_tb_score *sc = NULL){
	const int lane = (int)(threadIdx.x & 63);
	constexpr int C4 = (C + 3) / 4;
	constexpr int NLW = (128 / C) < NL ? (128 / C) : NL;     /* lanes of a row inside the window */
	constexpr int NDW = NLW * C4, ROWB = NLW * C;
	static_assert(NDW <= 64 && ROWB <= 128, "window geometry");
	uint32_t *S32 = (uint32_t*)tb; uint8_t *S8 = (uint8_t*)tb; uint8_t *Sd = S8 + 8192;
	int32_t i_ = x.qe, j_ = x.te; uint32_t d_ = 0;
#ifdef WTZ_EXP_NOTB
	i_ = -1; j_ = -1;      /* diagnostic build: no traceback (results are wrong) */
#endif
	uint32_t run_op = 0xFFu, run_len = 0;
	int32_t n_gap_runs = 0; bool consistent = true;
	wtz_cigw_t Wr; Wr.v = &cigars; Wr.tail = 0;
	int32_t cc = 0;
	if(i_ >= 0) cc = j_ - wtz_as_global(zb)[i_];
	const int ln_off = lane / C4, q4 = lane % C4;
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
			/* eight rows' loads are issued before the first is used; rows above row 0 re-read row 0 (never walked), idle lanes dword 0 (never stored) */
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
					/* the kernel's 5 decision bits -> the walker's byte: bits 1:0 move from H, bits 3:2 from E, bits 5:4 from F, bit 7 bases equal */
					const uint32_t w = w8[u];
					uint32_t v;
					if constexpr(EQ){
						const uint32_t a = (w >> 3) & 0x01010101u, b = (w >> 4) & 0x01010101u;
						v = (a << 1) | (b & (a ^ 0x01010101u)) | (w & 0x04040404u) | ((w & 0x02020202u) << 4) | ((w & 0x01010101u) << 7);
					} else {
						const uint32_t a = (w >> 2) & 0x01010101u, b = (w >> 3) & 0x01010101u;
						v = (a << 1) | (b & (a ^ 0x01010101u)) | ((w & 0x02020202u) << 1) | ((w & 0x01010101u) << 5);
					}
					const uint32_t pos = (uint32_t)(r8 + u) * 128u + pos0;
					if constexpr((C & 3) == 0) S32[pos >> 2] = v;
					else {
						#pragma unroll
						for(int k = 0; k < 4; k++) if(q4 * 4 + k < C) S8[pos + k] = (uint8_t)(v >> (8 * k));
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
				if((uint32_t)cc < (uint32_t)(NL * C)){          /* outside the band the kernel stored nothing: a zero byte, as the byte-wise staging had it */
					const int32_t t = cc - CC0;
					if((uint32_t)t >= (uint32_t)ROWB) break;
					zv = S8[rr * 128 + t];
				}
				const int32_t sft = (int32_t)Sd[rr];
				d_ = (zv >> (d_ << 1)) & 0x03;
				if(d_ == 0){ if(!EQ || (zv & 0x80u)) x.mat++; else x.mis++; i_--; j_--; cc += sft - 1; }      /* !EQ: x.mat counts the diagonal steps until the end */
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
		if constexpr(!EQ){
			const int32_t nd = x.mat, MXd = sc->M - sc->X;
			const long long S = (long long)x.score - sc->init - (long long)n_gap_runs * sc->O - (long long)sc->E * (x.ins + x.del) - (long long)sc->X * nd;
			if(MXd == 0 || S % MXd != 0 || S / MXd < 0 || S / MXd > nd) consistent = false;
			else { x.mat = (int32_t)(S / MXd); x.mis = nd - x.mat; }
		}
		x.aln = x.mat + x.mis + x.ins + x.del; x.qe++; x.te++;
	}
	return __builtin_amdgcn_readfirstlane((int)consistent) != 0;
}

/*
 * K-sw3 with the DP rows entirely in registers (the form the job kernel runs; wtz_extend_shift_wave_rt above is its
 * fallback and on-device cross-check).  Lane l owns the C band-relative columns l*C .. l*C+C-1.  The band start moves
 * by s = 0, 1 or 2 columns per row (kswx.h:186-199), so H(i-1,j-1) and E(i-1,j) of the new frame are the lane's own
 * registers at a compile-time offset (k+s-1, k+s) plus at most two values of the next lane / one of the previous lane,
 * fetched with wave_shl / wave_shr DPP moves: three fully unrolled row bodies, no LDS hand-over.  Cells are branch-free;
 * the row maximum and its FIRST arg-max come from one max-reduction over keys h*2048 + (2047 - band column) (callers
 * guarantee |h| < 2^20); the query row base is scalar (32-base words in VGPRs, v_readlane every 16 rows); LDS holds only
 * the 2-bit target.  Trace bytes, band starts and traceback are those of the rt form.
 */
template<int V> struct wtz_ic { static constexpr int value = V; };
template<int I, int N, typename F> WTZ_D void wtz_static_for(F &&f){ if constexpr(I < N){ f(wtz_ic<I>{}); wtz_static_for<I + 1, N>(f); } }
/* a*b + c with 24-bit a, b in ONE VALU op; the asm keeps the compiler from turning a 0/1 factor into compare+select or from
 * hoisting the (wave-uniform) product into a scalar register per cell */
WTZ_D int32_t wtz_mad24(int32_t a, int32_t b, int32_t c){ int32_t r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
template<int K>
WTZ_D int32_t wtz_mad24_imm(int32_t b, int32_t c){ int32_t r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "n"(K), "v"(b), "v"(c)); return r; }
template<int CMAX, int S>
WTZ_D void wtz_shift_row_inputs(int32_t (&hv)[CMAX], int32_t (&ev)[CMAX], int lane, int32_t bnd){
	/* rewrite hv := H(i-1, j-1), ev := E(i-1, j) for the new frame j = j0_old + S + k, in place */
	if(S == 0){
		int32_t prv = wtz_dpp_wave_shr1(-10000, hv[CMAX - 1]);
		prv = (lane == 0) ? bnd : prv;
		#pragma unroll
		for(int k = CMAX - 1; k > 0; k--) hv[k] = hv[k - 1];
		hv[0] = prv;
	} else if(S == 1){
		const int32_t ne0 = wtz_dpp_wave_shl1(-10000, ev[0]);
		#pragma unroll
		for(int k = 0; k + 1 < CMAX; k++) ev[k] = ev[k + 1];
		ev[CMAX - 1] = ne0;
	} else {
		const int32_t nh0 = wtz_dpp_wave_shl1(-10000, hv[0]);
		const int32_t ne0 = wtz_dpp_wave_shl1(-10000, ev[0]), ne1 = wtz_dpp_wave_shl1(-10000, ev[1]);
		#pragma unroll
		for(int k = 0; k + 1 < CMAX; k++) hv[k] = hv[k + 1];
		hv[CMAX - 1] = nh0;
		#pragma unroll
		for(int k = 0; k + 2 < CMAX; k++) ev[k] = ev[k + 2];
		ev[CMAX - 2] = ne0; ev[CMAX - 1] = ne1;
	}
}

template<int CMAX>
WTZ_D wtz_aln_t wtz_extend_shift_reg(int32_t qlen, const wtz_seq_packed &query, int32_t tlen, const wtz_seq_packed &target, int32_t init_score,
		int32_t ql, int32_t tl, int32_t W, int32_t M, int32_t X, int32_t I, int32_t D, int32_t E, int32_t T,
		uint64_t *tb, wtz_trace_t &tr, wtz_pool_t *pool, wtz_cigar_t &cigars, unsigned long long *cells, bool *ok){
	const int lane = (int)(threadIdx.x & 63);
	wtz_aln_t x; memset(&x, 0, sizeof x);
	*ok = true;
	if(lane == 0) cigars.n = 0;
	if(init_score < 0) init_score = 0;
	constexpr int C = CMAX;                       /* the lane block is exactly CMAX wide: 64*CMAX >= n_col */
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
	int32_t hv[C], ev[C];
	#pragma unroll
	for(int k = 0; k < C; k++){ hv[k] = -10000; ev[k] = -10000; }
	int32_t mx = init_score, mi = -1, mj = -1, gmax = 0, gi = -1, gj = -1;
	int32_t jbp = 0, c = 0, i;
	unsigned long long ncell = 0;
	const int32_t CE = C * E, IE = I + E, DE = D + E;
	uint32_t qw_lo = 0, qw_hi = 0, qcur = 0;
	const int32_t colrel0 = lane * C;
	int32_t jb_n = 0, je_n = tl; uint64_t tbits_n;
	{
		if(je_n > W + 1) je_n = W + 1;              /* row 0: c = 0 */
		if(je_n > tl) je_n = tl;
		const int32_t jj = colrel0 < tl ? colrel0 : (tl > 0 ? tl - 1 : 0);
		const int32_t w = jj >> 5, sh = (jj & 31) * 2;
		const uint64_t w0 = tb[w], w1 = tb[w + 1];
		tbits_n = sh ? ((w0 >> sh) | (w1 << (64 - sh))) : w0;
	}
	__builtin_amdgcn_s_waitcnt(0x0F70);              /* vmcnt(0) before the loop: a load still pending at loop entry would otherwise be waited for inside every row */
	for(i = 0; i < ql; i++){
		if((i & 63) == 0){
			/* Every branch of this block is wave-uniform BY CONSTRUCTION (readfirstlane) and the block ends in an explicit vmcnt(0): the
			 * wait-count pass works on the structurised CFG, and one vector load it believes may still be in flight when the row body
			 * starts makes every row wait for vmcnt(0) -- i.e. for the previous row's trace stores, the latency this kernel hides. */
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
			__builtin_amdgcn_s_waitcnt(0x0F70);          /* vmcnt(0), once per 64 rows */
		}
		/* band and target bases of THIS row were prepared at the end of the previous iteration (see below) */
		const int32_t jb = jb_n, je = je_n;
		if((i & 15) == 0){
			const int32_t qs = __builtin_amdgcn_readfirstlane((i & 2047) >> 5);
			qcur = (i & 16) ? (uint32_t)__builtin_amdgcn_readlane((int)qw_hi, qs) : (uint32_t)__builtin_amdgcn_readlane((int)qw_lo, qs);
		}
		const uint32_t qbase = (qcur >> ((i & 15) * 2)) & 3u;
		const int32_t j0 = jb + colrel0;
		const uint64_t tbits = tbits_n;
		/* ---- previous row into the new frame ---- */
		if(i == 0){
			#pragma unroll
			for(int k = 0; k < C; k++){ const int32_t j = j0 + k; hv[k] = (j == 0) ? init_score : init_score + D + E * j; }     /* rh[] initialisation, kswx.h:143-144; E stays -10000 */
		} else {
			const int32_t s = jb - jbp;
			const int32_t bnd = (jb == 0) ? init_score + I + E * i : -10000;      /* H(i-1, jb-1): outside the previous band unless it is column -1 */
			if(s == 0) wtz_shift_row_inputs<C, 0>(hv, ev, lane, bnd);
			else if(s == 1) wtz_shift_row_inputs<C, 1>(hv, ev, lane, bnd);
			else wtz_shift_row_inputs<C, 2>(hv, ev, lane, bnd);
		}
		/* The cells are instruction-issue bound (a lone wave retires one VALU op per 4 cycles), so every op per cell counts:
		 *  - base equality comes from one 64-bit XOR per row (eqw: bit 2k set where query base == target base k), not a compare per cell;
		 *  - the lane's F aggregate needs no validity mask: the cells right of the band end only feed lanes right of the band end;
		 *  - the four decisions are pushed into the trace byte as the sign bit of a difference (v_sub + v_alignbit), no compare/select;
		 *  - the arg-max key is taken from the masked H (the stored -10000 never wins) and the lane's column offset is added once.
		 * Trace byte of this kernel (decoded when the traceback stages it): bit 4 m<e, bit 3 max(m,e)<f, bit 2 E extended, bit 1 F extended, bit 0 bases equal. */
		const int32_t nv = je - j0;                        /* cell k of this lane is inside the band iff k < nv */
		uint32_t eq_lo, eq_hi;
		{
			const uint32_t qrep = 0x55555555u * qbase;
			const uint32_t x_lo = (uint32_t)tbits ^ qrep, x_hi = (uint32_t)(tbits >> 32) ^ qrep;
			eq_lo = ~(x_lo | (x_lo >> 1)) & 0x55555555u; eq_hi = ~(x_hi | (x_hi >> 1)) & 0x55555555u;
		}
		const int32_t MX = M - X, nE = -E;
		/* ---- m in place, the lane's F aggregate ---- */
		int32_t agg = -0x3FFFFFFF;
		wtz_static_for<0, C>([&](auto kc){
			constexpr int k = decltype(kc)::value;
			const int32_t b = (int32_t)(((k < 16 ? eq_lo : eq_hi) >> (2 * (k & 15))) & 1u);
			const int32_t m = wtz_mad24(b, MX, hv[k]) + X;
			hv[k] = m;
			const int32_t cand = wtz_mad24_imm<k>(nE, m);          /* m - k*E: the common DE + (C-1)*E is added after the loop */
			agg = cand > agg ? cand : agg;
		});
		agg += DE + (C - 1) * E;
		int32_t f;
		{
			const int32_t g = agg - lane * CE;
			const int32_t pm = wtz_wave_max_scan_excl(g, -0x3FFFFFFF);
			const int32_t from_prev = (lane == 0) ? -0x3FFFFFFF : pm + (lane - 1) * CE;
			const int32_t from_init = -10000 + lane * CE;
			f = from_prev > from_init ? from_prev : from_init;
		}
		/* ---- H, E', F, trace byte ---- */
		int32_t key = -0x40000000, kg = -0x40000000;
		uint32_t zw[C4];
		#pragma unroll
		for(int q4 = 0; q4 < C4; q4++) zw[q4] = 0;
		#pragma unroll
		for(int k = 0; k < C; k++){
			const bool valid = (k < nv);
			const int32_t m = hv[k], e = ev[k];
			const int32_t h0 = m > e ? m : e;
			uint32_t d = (uint32_t)(m - e) >> 31;                                              /* m < e */
			d = __builtin_amdgcn_alignbit(d, (uint32_t)(h0 - f), 31);                          /* max(m,e) < f */
			const int32_t h = h0 > f ? h0 : f;
			const int32_t te = m + IE, e2 = e + E;
			d = __builtin_amdgcn_alignbit(d, (uint32_t)(te - e2), 31);                         /* e + E > m + I + E */
			const int32_t en = e2 > te ? e2 : te;
			const int32_t tf = m + DE, f2 = f + E;
			d = __builtin_amdgcn_alignbit(d, (uint32_t)(tf - f2), 31);                         /* f + E > m + D + E */
			f = f2 > tf ? f2 : tf;
			d = (d << 1) | (((k < 16 ? eq_lo : eq_hi) >> (2 * (k & 15))) & 1u);
			const int32_t hm = valid ? h : -10000;
			hv[k] = hm; ev[k] = valid ? en : -10000;
			/* arg-max key h*2048 + (2047 - column): inside a group of 16 cells the column term is an inline constant of v_lshl_add_u32 */
			const int32_t kk = (int32_t)(((uint32_t)hm << 11) + (uint32_t)(-(k & 15)));
			kg = kk > kg ? kk : kg;
			if((k & 15) == 15 || k == C - 1){ const int32_t t = kg - (k & ~15); key = t > key ? t : key; kg = -0x40000000; }
			zw[k >> 2] |= (valid ? d : 0u) << (8 * (k & 3));
		}
		key += 2047 - colrel0;
		ncell += (unsigned long long)(je - jb);
		key = wtz_wave_max_i32(key);
		int32_t imax = 0, mj2 = -1;
		if((key >> 11) > 0){ imax = key >> 11; mj2 = jb + (2047 - (key & 2047)); }       /* first j with the maximum, only if > 0 (kswx.h:172) */
		if(lane == 0) wtz_as_global(zb)[i] = jb;
		if(je == tlen){
			const int32_t idx = je - 1 - jb, kl = idx % C;
			int32_t hsel = hv[0];
			#pragma unroll
			for(int k = 1; k < C; k++) hsel = (kl == k) ? hv[k] : hsel;
			const int32_t h1 = __builtin_amdgcn_readlane(hsel, __builtin_amdgcn_readfirstlane(idx / C));      /* H(i, je-1) */
			if(gmax < h1){ gmax = h1; gi = i; gj = je - 1; }
		}
		if(i + 1 == qlen && gmax < imax){ gmax = imax; gi = i; gj = mj2; }
		jbp = jb;
		bool stop = false;
		if(imax > mx){ mx = imax; mi = i; mj = mj2; }
		else if(imax <= 0) stop = true;
		if(!stop){
			c++; if(c < mj2) c++; else if(c > mj2) c--;
			/* next row's band and its target bases (LDS) BEFORE this row's trace goes out: the wait the compiler puts in front of
			 * an LDS read then only covers the stores of the previous row, which have had a whole row to complete */
			jb_n = 0; je_n = tl;
			if(jb_n < c - W) jb_n = c - W;
			if(je_n > c + W + 1) je_n = c + W + 1;
			if(je_n > tl) je_n = tl;
			const int32_t j0n = jb_n + colrel0;
			const int32_t jj = j0n < tl ? j0n : (tl > 0 ? tl - 1 : 0);
			const int32_t w = jj >> 5, sh = (jj & 31) * 2;
			const uint64_t w0 = tb[w], w1 = tb[w + 1];
			tbits_n = sh ? ((w0 >> sh) | (w1 << (64 - sh))) : w0;
		}
		{
			WTZ_GLOBAL_AS uint32_t *zr = wtz_as_global((uint32_t*)(z + (size_t)(i & 63) * zrow) + lane);
#ifndef WTZ_EXP_NOTRACE
			/* only the lanes that own band cells store: the bytes right of the band end are never read by the walk (E and H are -10000 there,
			 * no path enters them), and in the first rows of an extension half of the lane blocks lie beyond it - the trace stores were a
			 * quarter of this kernel's time (diagnostic builds without them: K-sw3 stage 927 -> 702 ms at configs[2]) */
			#pragma unroll
			for(int q4 = 0; q4 < C4; q4++) if(q4 * 4 < nv) zr[(size_t)q4 * 64] = zw[q4];
#else
			if(zw[0] == 0xFFFFFFFFu && zw[C4 - 1] == 0xFFFFFFFEu) zr[0] = 1;     /* diagnostic build: never true, keeps the trace computation alive */
#endif
		}
		if(stop) break;
	}
	if(cells && lane == 0) *cells += ncell;
	if(!*ok) return x;
	if(gmax > 0 && gmax >= mx + T){ x.score = gmax; x.qe = gi; x.te = gj; }
	else { x.score = mx; x.qe = mi; x.te = mj; }
	__threadfence_block();     /* the trace was written by lanes of THIS wave: ordering inside the wave is enough (an agent-scope fence would write back the whole L2) */
	const unsigned long long pt_tb3 = WTZ_PROF_T(); (void)pt_tb3;
	WTZ_PROF_CNT(60, i < ql ? i + 1 : ql); WTZ_PROF_CNT(61, 1);
	wtz_shift_traceback<C, 64>(x, zchunk, zb, zrow, tb, cigars);
	WTZ_PROF_ADD(9, pt_tb3);
	return wtz_bcast_aln(x);
}

/* ---- K-sw3 jobs: one wave (64 threads) per job; jobs that do not fit the LDS rings run the scalar body on lane 0 ---- */
template<int P, int TW>
__global__ void __launch_bounds__(64) wtz_kernel_extjobs(wtz_extjob_t *jobs, const uint32_t *order, uint32_t n, const wtz_params_t *Pm, wtz_pool_t *pool, wtz_pool_t *tpool){
	__shared__ int32_t sHs[P]; __shared__ int32_t sEs[P]; __shared__ uint64_t stb[TW];
	const uint32_t b = blockIdx.x;
	if(b >= n) return;
	wtz_extjob_t *job = &jobs[order ? order[b] : b];
	if(!job->valid || job->done) return;
	const int lane = (int)(threadIdx.x & 63);
	int32_t init_score = job->init_score < 0 ? 0 : job->init_score;
	if(job->qlen <= 0 || job->tlen <= 0){
		if(lane == 0){ wtz_aln_t x; memset(&x, 0, sizeof x); x.score = init_score; job->x = x; job->cigar = NULL; job->cigar_len = 0; job->bad = 0; job->cells = 0; }
		return;
	}
	wtz_wave_lds_t L; L.Hs = sHs; L.Es = sEs; L.tb = stb; L.PM = P - 1; L.tw = TW;
	int32_t W = job->W, ql, tl, n_col;
	wtz_ext_geometry(job->qlen, job->tlen, init_score, W, Pm->M, Pm->O, Pm->O, Pm->E, Pm->T, ql, tl, n_col);
	if(wtz_wave_fits(L, n_col, tl, ql)){
		wtz_trace_t tr; tr.chunk = NULL; tr.zb = NULL; tr.n_chunk = 0; tr.zrow = 0; tr.cap_rows = 0;
		wtz_cigar_t cg; if(lane == 0) cg.init(pool, 64);
		unsigned long long cells = 0; bool ok = true;
		const int32_t Cw = (((n_col + 63) / 64) | 1);
		wtz_aln_t x;
		if(Cw <= 8)       x = wtz_extend_shift_wave_rt<8>(job->qlen, job->q, job->tlen, job->t, job->init_score, job->W, Pm->M, Pm->X, Pm->O, Pm->O, Pm->E, Pm->T, L, tr, tpool, cg, &cells, &ok);
		else if(Cw <= 16) x = wtz_extend_shift_wave_rt<16>(job->qlen, job->q, job->tlen, job->t, job->init_score, job->W, Pm->M, Pm->X, Pm->O, Pm->O, Pm->E, Pm->T, L, tr, tpool, cg, &cells, &ok);
		else              x = wtz_extend_shift_wave_rt<32>(job->qlen, job->q, job->tlen, job->t, job->init_score, job->W, Pm->M, Pm->X, Pm->O, Pm->O, Pm->E, Pm->T, L, tr, tpool, cg, &cells, &ok);
		if(lane == 0){ job->x = x; job->cigar = cg.a; job->cigar_len = cg.n; job->bad = (!ok || cg.bad); job->cells = cells; job->done = 3; }
	} else if(lane == 0){
		wtz_swmem_t mem; wtz_swmem_init(mem, tpool);
		wtz_cigar_t cg; cg.init(pool, 64);
		unsigned long long cells = 0;
		wtz_aln_t x = wtz_extend_shift(job->qlen, job->q, job->tlen, job->t, job->init_score, job->W, Pm->M, Pm->X, Pm->O, Pm->O, Pm->E, Pm->T, mem, cg, &cells);
		job->x = x; job->cigar = cg.a; job->cigar_len = cg.n; job->bad = (mem.bad || cg.bad); job->cells = cells; job->done = 3;
	}
}

/* K-sw3 jobs through the register DP: LDS carries only the 2-bit target.  Jobs outside its envelope (band wider than
 * 64*32 columns, target longer than the LDS words, scores beyond the packed-key range) are left for wtz_kernel_extjobs. */
/* CLO < columns per lane <= CHI: the whole envelope is <TW, 0, 32>; the launch may be split by band class (run_extjobs) so that the narrow
 * bands run from a kernel with a smaller register budget and a smaller instruction footprint */
template<int TW, int CLO = 0, int CHI = 32>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CHI <= 16 ? 2 : WTZ_OCC_EXTREG, 8))) wtz_kernel_extjobs_reg(wtz_extjob_t *jobs, const uint32_t *order, uint32_t n, const wtz_params_t *Pm, wtz_pool_t *pool, wtz_pool_t *tpool){
	__shared__ uint64_t stb[TW];
	const uint32_t b = blockIdx.x;
	if(b >= n) return;
	wtz_extjob_t *job = &jobs[order ? order[b] : b];
	if(!job->valid) return;
	const int lane = (int)(threadIdx.x & 63);
	if(job->qlen <= 0 || job->tlen <= 0) return;
	const unsigned long long pt_job = WTZ_PROF_T(); (void)pt_job;
	const int32_t init_score = job->init_score < 0 ? 0 : job->init_score;
	int32_t W = job->W, ql, tl, n_col;
	wtz_ext_geometry(job->qlen, job->tlen, init_score, W, Pm->M, Pm->O, Pm->O, Pm->E, Pm->T, ql, tl, n_col);
	const int32_t Cw = (n_col + 63) / 64;
	if(Cw > 32 || (tl + 63) / 32 + 1 > TW || (ql + 63) / 64 > WTZ_TRACE_MAXCHUNK) return;
	if((long long)init_score + (long long)Pm->M * (ql < tl ? ql : tl) >= (1 << 20)) return;
	if(Cw <= CLO || (CHI < 32 && Cw > CHI)) return;        /* another launch's band class */
	WTZ_PROF_BEGIN();
	wtz_trace_t tr; tr.chunk = NULL; tr.zb = NULL; tr.n_chunk = 0; tr.zrow = 0; tr.cap_rows = 0;
	wtz_cigar_t cg; cg.a = NULL; cg.n = cg.cap = 0; cg.pool = pool; cg.bad = 0;
	if(lane == 0) cg.init(pool, (uint32_t)ql / 2u + 16u);
	unsigned long long cells = 0; bool ok = true;
	wtz_aln_t x;
#define WTZ_EXTREG_CASE(CM) x = wtz_extend_shift_reg<CM>(job->qlen, job->q, job->tlen, job->t, job->init_score, ql, tl, W, Pm->M, Pm->X, Pm->O, Pm->O, Pm->E, Pm->T, stb, tr, tpool, cg, &cells, &ok)
	if(Cw <= 4){ if(CLO < 4 && CHI >= 4) WTZ_EXTREG_CASE(4); }
	else if(Cw <= 8){ if(CLO < 8 && CHI >= 8) WTZ_EXTREG_CASE(8); }
	else if(Cw <= 12){ if(CLO < 12 && CHI >= 12) WTZ_EXTREG_CASE(12); }
	else if(Cw <= 16){ if(CLO < 16 && CHI >= 16) WTZ_EXTREG_CASE(16); }
	else if(Cw <= 20){ if(CLO < 20 && CHI >= 20) WTZ_EXTREG_CASE(20); }
	else if(Cw <= 24){ if(CLO < 24 && CHI >= 24) WTZ_EXTREG_CASE(24); }
	else if(Cw <= 28){ if(CLO < 28 && CHI >= 28) WTZ_EXTREG_CASE(28); }
	else { if(CHI >= 32) WTZ_EXTREG_CASE(32); }
#undef WTZ_EXTREG_CASE
	if(lane == 0){ job->x = x; job->cigar = cg.a; job->cigar_len = cg.n; job->bad = (!ok || cg.bad); job->cells = cells; job->done = 1; }
	WTZ_PROF_ADD(8, pt_job); WTZ_PROF_MAX(15, pt_job); WTZ_PROF_CNT(10, 1);
	WTZ_PROF_END();
}

/* four int32 as one 128-bit LDS access, and the workgroup barrier of the multi-wave K-sw3 form (wtz_sw_frame_mw.h): the row's trace stores are NOT waited for */
struct alignas(16) wtz_i4 { int32_t v[4]; };

WTZ_D void wtz_mw_barrier(){ asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

/* (the round-4 four-wave kernel wtz_extend_shift_mw / wtz_kernel_extjobs_mw lived here: retired in round 6 - its frame-form successor is wtz_sw_frame_mw.h, DP form 6) */

/*
 * K-sw1 for the small problems between two anchors of a window (80 % have a band of <= 64 columns, 96 % <= 128 rows):
 * the whole DP state lives in registers.  Lane l owns the C band-relative columns l*C .. l*C+C-1; the fixed band moves
 * right by exactly one column per row once i > W, so the hand-over between rows is a register rotation plus one
 * wave_shl DPP per array (no LDS rings).  The query is staged as 32-base words in VGPRs (the row's base comes from a
 * v_readlane), the target as 32-base words in LDS; the trace is kept in LDS at 4 bits per cell (two rows per byte), so
 * the traceback of lane 0 never leaves the CU.  Results are those of wtz_extend_wave<1> / kswx_extend_align_core.
 * Requirements (checked by the caller): n_col <= 64*C, ((ql+1)/2)*64*C <= ztr_bytes, (tl+63)/32+1 <= tb words, ql <= 2048.
 */

/* one step of the K-sw1 / kswx traceback on lane 0, written without branches (the branchy form spent its time in exec-mask
 * bookkeeping and taken branches, not in arithmetic): next state from the cell's nibble, base equality from the 2-bit words of
 * both sequences in LDS (read whether the step is diagonal or not), counters and cursor by flag arithmetic; the open run is kept
 * in runs[nr] and simply rewritten every step. */
typedef struct { int32_t i, j; uint32_t d, run_op, run_len, nr; int32_t ndiag, mis, i0, j0; } wtz_walk_t;      /* i0, j0: where the walk started */
WTZ_D void wtz_walk_step(wtz_walk_t &w, uint32_t nib, const uint32_t *qw32, const uint32_t *tb32, uint32_t *runs){
	const uint32_t h3 = nib & 3u, h = h3 > 2u ? 2u : h3;
	const uint32_t d = ((h | (nib & 4u) | ((nib & 8u) << 2)) >> (w.d << 1)) & 3u;      /* bits 1:0 move from H, 3:2 from E (0 / 1), 5:4 from F (0 / 2) */
	const uint32_t qb = (qw32[w.i >> 4] >> ((w.i & 15) * 2)) & 3u, tq = (tb32[w.j >> 4] >> ((w.j & 15) * 2)) & 3u;
	const uint32_t isd = d == 0 ? 1u : 0u, ne = qb != tq ? 1u : 0u;
	w.ndiag += (int32_t)isd; w.mis += (int32_t)(isd & ne);          /* insertions / deletions follow from the distance walked (wtz_walk_finish) */
	w.i -= d != 2 ? 1 : 0; w.j -= d != 1 ? 1 : 0;
	const bool same = d == w.run_op;
	w.nr += (!same && w.run_len) ? 1u : 0u;
	w.run_len = same ? w.run_len + 1u : 1u; w.run_op = d; w.d = d;
	runs[w.nr] = (w.run_len << 4) | d;
}
/* the two leading gaps (kswx.h:321-322) merge with the open run when the operation agrees (kswx_push_cigar); closes the list */
WTZ_D void wtz_walk_finish(wtz_walk_t &w, wtz_aln_t &x, uint32_t *runs, uint32_t *n_runs){
	uint32_t run_op = w.run_op, run_len = w.run_len, nr = w.nr;
	x.mat = w.ndiag - w.mis; x.mis = w.mis; x.ins = (w.i0 - w.i) - w.ndiag; x.del = (w.j0 - w.j) - w.ndiag;      /* every step but a deletion moves up, every step but an insertion moves left */
	if(w.i >= 0){ x.ins += w.i + 1; if(run_len && run_op == 1u) run_len += (uint32_t)(w.i + 1); else { if(run_len) runs[nr++] = (run_len << 4) | run_op; run_op = 1u; run_len = (uint32_t)(w.i + 1); } }
	if(w.j >= 0){ x.del += w.j + 1; if(run_len && run_op == 2u) run_len += (uint32_t)(w.j + 1); else { if(run_len) runs[nr++] = (run_len << 4) | run_op; run_op = 2u; run_len = (uint32_t)(w.j + 1); } }
	if(run_len) runs[nr++] = (run_len << 4) | run_op;
	*n_runs = nr;                              /* in traceback order: the caller replays them backwards */
	x.aln = x.mat + x.mis + x.ins + x.del; x.qe++; x.te++;
}

/* ZG: the trace does not fit LDS and lives in the pool (HBM): the rows store it through a global-address-space pointer (one
 * contiguous zrow-byte piece per row pair), the traceback walks LDS-staged blocks of 64 DP rows x the whole band (`stage`, 4 KB) */
template<int C, bool ZG = false>
WTZ_D wtz_aln_t wtz_extend_fixed_reg(int32_t qlen, const wtz_seq_packed &query, int32_t tlen, const wtz_seq_packed &target, int32_t init_score,
		int32_t ql, int32_t tl, int32_t W, int32_t M, int32_t X, int32_t I, int32_t D, int32_t E, int32_t T,
		uint64_t *tb, uint32_t *qlds, uint8_t *ztr, uint32_t zrow, uint32_t *runs, uint32_t *n_runs, unsigned long long *cells, uint8_t *stage = NULL){
	const int lane = (int)(threadIdx.x & 63);
	constexpr int KB = C <= 2 ? 7 : 9;           /* bits of the band column inside the packed arg-max key (callers check |h| < 2^(31-KB)) */
	wtz_aln_t x; memset(&x, 0, sizeof x);
	*n_runs = 0;
	if(init_score < 0) init_score = 0;
	const unsigned long long pt_stage = WTZ_PROF_T();
	/* ---- stage both sequences ---- */
	{
		const int32_t nw = (tl + 31) / 32 + 1;
		for(int32_t w = lane; w < nw; w += 64) tb[w] = wtz_pack32(target, w * 32, tl);
	}
	const uint64_t qw = wtz_pack32(query, lane * 32, ql);
	const uint32_t qw_lo = (uint32_t)qw, qw_hi = (uint32_t)(qw >> 32);
	((uint64_t*)qlds)[lane] = qw;                    /* for the traceback: base i is bits 2*(i&15) of word i>>4 */
	__threadfence_block();
	WTZ_PROF_ADD(5, pt_stage);
	const unsigned long long pt_rows = WTZ_PROF_T();
	/* The row loop is written for instruction count: every cell update is branch-free (max / compare-select), the row
	 * maximum and its LAST arg-max come out of ONE integer max-reduction over packed keys h*128 + band column (the caller
	 * guarantees |h| < 2^23), the lane's target bases sit in a 16-base register window that is shifted once per row and
	 * refilled from LDS every 16-C rows, and the row's query base is scalar. */
	int32_t hp[C], ep[C]; uint32_t nibp[C];
	#pragma unroll
	for(int k = 0; k < C; k++){ hp[k] = -10000; ep[k] = -10000; nibp[k] = 0; }
	int32_t mx = init_score, mi = -1, mj = -1, gmax = 0, gi = -1, gj = -1;
	int32_t jbp = 0, i, i_done = -1;
	unsigned long long ncell = 0;
	const int32_t CE = C * E, IE = I + E, DE = D + E;
	const uint32_t *tb32 = (const uint32_t*)tb;
	uint32_t tw = 0; int32_t tw_left = 0; uint32_t qcur = 0;
	const int32_t colrel0 = lane * C;
	for(i = 0; i < ql; i++){
		int32_t jb = i - W; if(jb < 0) jb = 0;
		int32_t je = i + W + 1; if(je > tl) je = tl;
		if((i & 15) == 0){
			const int32_t qs = __builtin_amdgcn_readfirstlane(i >> 5);
			qcur = (i & 16) ? (uint32_t)__builtin_amdgcn_readlane((int)qw_hi, qs) : (uint32_t)__builtin_amdgcn_readlane((int)qw_lo, qs);
		}
		const uint32_t qbase = (qcur >> ((i & 15) * 2)) & 3u;
		const int32_t j0 = jb + colrel0;
		const bool moved = (i > 0) && (jb != jbp);
		if(moved){ tw >>= 2; tw_left--; }
		if(i == 0 || tw_left < C){
			const int32_t jj = j0 < tl ? j0 : (tl > 0 ? tl - 1 : 0);
			const uint32_t lo = tb32[jj >> 4], hi = tb32[(jj >> 4) + 1];
			tw = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(jj & 15) * 2u);
			tw_left = 16;
		}
		/* ---- predecessors from the previous row's registers ---- */
		int32_t pred[C], ein[C];
		if(i == 0){
			#pragma unroll
			for(int k = 0; k < C; k++){ const int32_t j = j0 + k; pred[k] = (j == 0) ? init_score : init_score + D + E * j; ein[k] = -10000; }     /* rh[] / re[] initialisation, kswx.h:143-146 */
		} else if(moved){                     /* band moved right by one: H(i-1,j-1) is the lane's own column, E(i-1,j) the next one */
			const int32_t nxt = wtz_dpp_wave_shl1(-10000, ep[0]);
			#pragma unroll
			for(int k = 0; k < C; k++){ pred[k] = hp[k]; ein[k] = (k + 1 < C) ? ep[k + 1] : nxt; }
		} else {                              /* band still starts at column 0 */
			int32_t prv = wtz_dpp_wave_shr1(-10000, hp[C - 1]);
			prv = (lane == 0) ? init_score + I + E * i : prv;           /* H(i-1,-1), kswx.h:262 */
			#pragma unroll
			for(int k = 0; k < C; k++){ pred[k] = k ? hp[k - 1] : prv; ein[k] = ep[k]; }
		}
		/* ---- m and the lane's F aggregate ---- */
		int32_t mv[C]; bool valid[C]; int32_t agg = -0x3FFFFFFF;
		#pragma unroll
		for(int k = 0; k < C; k++){
			const uint32_t tbase = (tw >> (2 * k)) & 3u;
			mv[k] = pred[k] + ((qbase == tbase) ? M : X);
			valid[k] = (j0 + k < je);
			const int32_t cand = mv[k] + DE + (C - 1 - k) * E;
			agg = (valid[k] && cand > agg) ? cand : agg;
		}
		int32_t f;
		{
			const int32_t g = agg - lane * CE;
			const int32_t pm = wtz_wave_max_scan_excl(g, -0x3FFFFFFF);
			const int32_t from_prev = (lane == 0) ? -0x3FFFFFFF : pm + (lane - 1) * CE;
			const int32_t from_init = -10000 + lane * CE;
			f = from_prev > from_init ? from_prev : from_init;
		}
		/* ---- H, E', F, trace nibble ---- */
		int32_t key = (int32_t)0x80000000;
		#pragma unroll
		for(int k = 0; k < C; k++){
			const int32_t m = mv[k], e = ein[k];
			const int32_t h0 = m > e ? m : e;
			const int32_t te = m + IE, e2 = e + E, tf = m + DE, f2 = f + E;
			/* the four decisions as sign bits of differences (|values| < 2^23: no overflow), shifted in with v_alignbit instead of four
			 * compare + select pairs: bit 0 m < e, bit 1 max(m,e) < f, bit 2 E extended, bit 3 F extended; the walker reads the move
			 * from H as min(bits 1:0, 2) */
			uint32_t nib = (uint32_t)(tf - f2) >> 31;
			nib = __builtin_amdgcn_alignbit(nib, (uint32_t)(te - e2), 31);
			nib = __builtin_amdgcn_alignbit(nib, (uint32_t)(h0 - f), 31);
			nib = __builtin_amdgcn_alignbit(nib, (uint32_t)(m - e), 31);
			const int32_t h = h0 > f ? h0 : f;
			const int32_t en = e2 > te ? e2 : te;
			f = f2 > tf ? f2 : tf;
			hp[k] = valid[k] ? h : -10000; ep[k] = valid[k] ? en : -10000;
			nib = valid[k] ? nib : 0u;
			const int32_t kk = h * (1 << KB) + (colrel0 + k);
			key = (valid[k] && kk > key) ? kk : key;
			if(i & 1){
				if((uint32_t)(colrel0 + k) < zrow){
					if(ZG) wtz_as_global(ztr)[(size_t)(i >> 1) * zrow + colrel0 + k] = (uint8_t)(nibp[k] | (nib << 4));
					else ztr[(size_t)(i >> 1) * zrow + colrel0 + k] = (uint8_t)(nibp[k] | (nib << 4));
				}
			}
			else nibp[k] = nib;
		}
		i_done = i;
		ncell += (unsigned long long)(je - jb);
		key = wtz_wave_max_i32(key);
		int32_t imax = 0, mj2 = -1;
		if((key >> KB) >= 0){ imax = key >> KB; mj2 = jb + (key & ((1 << KB) - 1)); }           /* last j with h >= running max >= 0, kswx.h:288-289 */
		if(je == tlen){
			const int32_t idx = je - 1 - jb;
			int32_t hsel = hp[0];
			#pragma unroll
			for(int k = 1; k < C; k++) hsel = (idx % C == k) ? hp[k] : hsel;
			const int32_t h1 = __builtin_amdgcn_readlane(hsel, __builtin_amdgcn_readfirstlane(idx / C));      /* H(i, je-1) */
			if(gmax < h1){ gmax = h1; gi = i; gj = je - 1; }
		}
		if(i + 1 == qlen && gmax < imax){ gmax = imax; gi = i; gj = mj2; }
		jbp = jb;
		if(imax > mx){ mx = imax; mi = i; mj = mj2; }
		else if(imax <= 0) break;
	}
	if(i_done >= 0 && !(i_done & 1)){          /* the last row was the first of its byte pair */
		#pragma unroll
		for(int k = 0; k < C; k++) if((uint32_t)(lane * C + k) < zrow) ztr[(size_t)(i_done >> 1) * zrow + lane * C + k] = (uint8_t)nibp[k];
	}
	if(cells && lane == 0) *cells += ncell;
	if(gmax > 0 && gmax >= mx + T){ x.score = gmax; x.qe = gi; x.te = gj; }
	else { x.score = mx; x.qe = mi; x.te = mj; }
	__threadfence_block();
	WTZ_PROF_ADD(2, pt_rows);
	WTZ_PROF_CNT(4, i_done + 1);
	const unsigned long long pt_tb = WTZ_PROF_T();
	wtz_walk_t wk; wk.i = x.qe; wk.j = x.te; wk.d = 0; wk.run_op = 0xFFu; wk.run_len = 0; wk.nr = 0; wk.ndiag = wk.mis = 0; wk.i0 = wk.i; wk.j0 = wk.j;
	if(ZG){
		uint32_t *stage32 = (uint32_t*)stage;
		const int32_t RB = zrow <= 128 ? 32 : (zrow <= 256 ? 16 : 8);         /* packed rows per staged block: RB * zrow <= 4 KB */
		int32_t i_ = wk.i, j_ = wk.j;
		while(i_ >= 0 && j_ >= 0){
			const int32_t p1 = i_ >> 1, p0 = p1 >= RB - 1 ? p1 - (RB - 1) : 0;
			{
				const uint32_t nd = (uint32_t)(p1 - p0 + 1) * (zrow >> 2);
				const uint32_t *src = (const uint32_t*)(ztr + (size_t)p0 * zrow);
				for(uint32_t xw = (uint32_t)lane; xw < nd; xw += 64) stage32[xw] = src[xw];
			}
			__threadfence_block();
			if(lane == 0){
				while((wk.i | wk.j) >= 0 && (wk.i >> 1) >= p0){
					const int32_t col = wk.j - (wk.i > W ? wk.i - W : 0);
					const uint32_t zv = stage[(size_t)((wk.i >> 1) - p0) * zrow + col];
					wtz_walk_step(wk, zv >> ((wk.i & 1) * 4), qlds, tb32, runs);
				}
			}
			i_ = __builtin_amdgcn_readfirstlane(wk.i); j_ = __builtin_amdgcn_readfirstlane(wk.j);
			__threadfence_block();
		}
		if(lane == 0) wtz_walk_finish(wk, x, runs, n_runs);
	} else
	if(lane == 0){
		while((wk.i | wk.j) >= 0){
			const int32_t col = wk.j - (wk.i > W ? wk.i - W : 0);
			const uint32_t zv = ztr[(size_t)(wk.i >> 1) * zrow + col];
			wtz_walk_step(wk, zv >> ((wk.i & 1) * 4), qlds, tb32, runs);
		}
		wtz_walk_finish(wk, x, runs, n_runs);
	}
	WTZ_PROF_ADD(3, pt_tb);
	return wtz_bcast_aln(x);
}

/* up to 64 bases of a view in two registers */
struct wtz_seq_reg2 { uint64_t w0, w1; WTZ_D uint32_t at(int32_t i) const { return (uint32_t)(((i < 32) ? (w0 >> (2 * i)) : (w1 >> (2 * (i - 32)))) & 3u); } };
/* hz_align_hzmo (hzm_aln.h:278-314) over register-resident sequences; W == NULL only scores.  A mismatching z-mer must leave
 * the CIGAR untouched: the caller snapshots the writer (open run + vector length) and restores it when aln == 0 */
template<typename S1, typename S2>
WTZ_D wtz_aln_t wtz_align_zmer_w(const S1 &pb1, uint32_t len1, const S2 &pb2, uint32_t len2, int32_t M, int32_t I, int32_t D, int32_t E, wtz_cigw_t *W){
	wtz_aln_t x, zero; memset(&zero, 0, sizeof zero); x = zero;
	uint32_t s0 = 0, s1 = 0, e0, e1, l0, l1;
	while(s0 < len1 || s1 < len2){
		const uint32_t b0 = pb1.at((int32_t)s0);
		if(b0 != pb2.at((int32_t)s1)) return zero;
		e0 = s0 + 1; while(e0 < len1 && pb1.at((int32_t)e0) == b0) e0++;
		e1 = s1 + 1; while(e1 < len2 && pb2.at((int32_t)e1) == b0) e1++;
		l0 = e0 - s0; l1 = e1 - s1;
		if(l0 < l1){
			x.aln += l1; x.mat += l0; x.ins += l1 - l0; x.score += (int32_t)l0 * M + I + (int32_t)(l1 - l0) * E;
			if(W){ wtz_cigw_push(*W, 0, l0); wtz_cigw_push(*W, 1, l1 - l0); }
		} else if(l0 == l1){
			x.aln += l0; x.mat += l0; x.score += (int32_t)l0 * M;
			if(W) wtz_cigw_push(*W, 0, l0);
		} else {
			x.aln += l0; x.mat += l1; x.del += l0 - l1; x.score += (int32_t)l1 * M + D + (int32_t)(l0 - l1) * E;
			if(W){ wtz_cigw_push(*W, 0, l1); wtz_cigw_push(*W, 2, l0 - l1); }
		}
		s0 = e0; s1 = e1;
	}
	x.te = x.mat + x.del; x.qe = x.mat + x.ins;
	return x;
}

/* ---- one K-sw1 problem on the wavefront: picks the device form from the problem's shape (register DP with 1 / 2 / 4 / 8 band columns
 *      per lane, 4-bit trace in the LDS slice or in the pool; the scalar body for what is outside every envelope) and runs it.
 *      TEST = true is the form of wtz_test_dp (function-level parity vectors): `force` then names the form instead
 *      (C | 16 = trace in the pool, 255 = scalar body) and R.form = -1 reports a problem outside the forced form's envelope.
 *      LDS slice: 128 target words (1 KB) at L.tb, then the 4-bit trace with its run list at the top end (ztr, ztr_bytes). ---- */
typedef struct { wtz_aln_t y; uint32_t *runs; uint32_t n_runs; bool lds_runs, ok, defer; int form; } wtz_fixres_t;
#define WTZ_FORM_SCALAR 255
template<bool FULL, bool TEST>
WTZ_D void wtz_fixed_problem_wave(int32_t qlen, const wtz_seq_packed &q, int32_t tlen, const wtz_seq_packed &t, int32_t score_in, const wtz_params_t *P,
		const wtz_wave_lds_t &L, uint8_t *ztr, int32_t ztr_bytes, wtz_cigar_t &tmp, wtz_swmem_t &mem, wtz_pool_t *pool, unsigned long long *cells, int force, wtz_fixres_t &R){
	const int lane = (int)(threadIdx.x & 63);
	const int32_t M = P->M, X = P->X, I = P->O, D = P->O, E = P->E, T = P->T;
	wtz_aln_t y; memset(&y, 0, sizeof y);
	R.runs = NULL; R.n_runs = 0; R.lds_runs = false; R.ok = true; R.defer = false; R.form = 0;
	const unsigned long long pt0 = WTZ_PROF_T(); (void)pt0;
	int32_t init = score_in < 0 ? 0 : score_in, W = P->w, ql = 0, tl = 0, n_col = 0; bool okk = true;
	if(qlen > 0 && tlen > 0) wtz_ext_geometry(qlen, tlen, init, W, M, I, D, E, T, ql, tl, n_col);
	const int32_t run_bytes = 4 * (ql + tl + 4);
	const int32_t zrow = (n_col + 3) & ~3;                 /* trace bytes per row pair: one nibble pair per band column */
	const int32_t hmax = init + M * (ql < tl ? ql : tl);                 /* bound of |h|: the register DP packs h and the band column into one int32 key */
	const bool shape = (qlen > 0 && tlen > 0 && ql <= 2048 && (tl + 63) / 32 + 1 <= L.tw);
	bool lds_fit = shape && n_col <= 128 && hmax < (1 << 23) && ((ql + 1) / 2) * zrow + run_bytes <= ztr_bytes;
	bool pool_fit = shape && !lds_fit && n_col <= 512 && hmax < (n_col <= 128 ? (1 << 23) : (1 << 21)) && 4096 + run_bytes <= ztr_bytes;
	int32_t cmin = n_col <= 64 ? 1 : (n_col <= 128 ? 2 : (n_col <= 256 ? 4 : 8));       /* band columns per lane */
	bool scalar = false;
	if constexpr(TEST){
		if(force == WTZ_FORM_SCALAR){ lds_fit = pool_fit = false; scalar = true; }
		else if(force){
			const int32_t fc = force & 15; const bool zg = (force & 16) != 0;
			const bool key_ok = hmax < (fc <= 2 ? (1 << 23) : (1 << 21));
			const bool can = shape && (fc == 1 || fc == 2 || fc == 4 || fc == 8) && fc >= cmin && n_col <= 64 * fc && key_ok && (FULL || fc <= 2)
				&& (zg ? (4096 + run_bytes <= ztr_bytes) : (((ql + 1) / 2) * zrow + run_bytes <= ztr_bytes && fc <= 2));
			if(!can && !(qlen <= 0 || tlen <= 0)){ R.form = -1; R.y = y; return; }
			lds_fit = can && !zg; pool_fit = can && zg; cmin = fc;
		}
	}
	if(!FULL && !(qlen <= 0 || tlen <= 0) && !(lds_fit || (pool_fit && n_col <= 128))){ R.defer = true; R.y = y; return; }
	if(lds_fit){
		R.runs = (uint32_t*)(ztr + ztr_bytes - run_bytes); R.lds_runs = true; R.form = cmin;
		if(cmin == 1) y = wtz_extend_fixed_reg<1>(qlen, q, tlen, t, score_in, ql, tl, W, M, X, I, D, E, T, L.tb, L.qw, ztr, (uint32_t)zrow, R.runs, &R.n_runs, cells);
		else          y = wtz_extend_fixed_reg<2>(qlen, q, tlen, t, score_in, ql, tl, W, M, X, I, D, E, T, L.tb, L.qw, ztr, (uint32_t)zrow, R.runs, &R.n_runs, cells);
	} else if(pool_fit){
		/* the 4-bit trace does not fit the LDS slice (or the band is wider than 128 columns): trace in the pool, traceback through a
		 * 4 KB LDS stage, run list in LDS above the stage */
		unsigned long long za = 0;
		if(lane == 0) za = (unsigned long long)(uintptr_t)wtz_pool_alloc(pool, (size_t)((ql + 1) / 2) * zrow);
		za = __shfl(za, 0, 64);
		if(za == 0){ R.ok = false; R.y = y; return; }
		uint8_t *zg = (uint8_t*)(uintptr_t)za;
		R.runs = (uint32_t*)(ztr + ztr_bytes - run_bytes); R.lds_runs = true; R.form = cmin | 16;
		if(cmin == 1)      y = wtz_extend_fixed_reg<1, true>(qlen, q, tlen, t, score_in, ql, tl, W, M, X, I, D, E, T, L.tb, L.qw, zg, (uint32_t)zrow, R.runs, &R.n_runs, cells, ztr);
		else if(cmin == 2) y = wtz_extend_fixed_reg<2, true>(qlen, q, tlen, t, score_in, ql, tl, W, M, X, I, D, E, T, L.tb, L.qw, zg, (uint32_t)zrow, R.runs, &R.n_runs, cells, ztr);
		else if constexpr(FULL){
			if(cmin == 4) y = wtz_extend_fixed_reg<4, true>(qlen, q, tlen, t, score_in, ql, tl, W, M, X, I, D, E, T, L.tb, L.qw, zg, (uint32_t)zrow, R.runs, &R.n_runs, cells, ztr);
			else          y = wtz_extend_fixed_reg<8, true>(qlen, q, tlen, t, score_in, ql, tl, W, M, X, I, D, E, T, L.tb, L.qw, zg, (uint32_t)zrow, R.runs, &R.n_runs, cells, ztr);
		}
		WTZ_PROF_ADD(55, pt0); WTZ_PROF_CNT(7, 1); WTZ_PROF_CNT(13, ql); WTZ_PROF_CNT(14, n_col);
	} else if(qlen <= 0 || tlen <= 0){
		/* empty problem (wtz_extend_fixed: score = init, nothing aligned, empty CIGAR) */
		memset(&y, 0, sizeof y); y.score = init; if(lane == 0) tmp.n = 0;
	} else if constexpr(FULL){
		/* whatever is outside the register DP's envelope (rows > 2048, band > 512 columns, huge scores): the scalar body */
		(void)scalar;
		const unsigned long long ptw = WTZ_PROF_T(); (void)ptw;
		R.form = WTZ_FORM_SCALAR;
		if(lane == 0){ tmp.n = 0; y = wtz_extend_fixed(qlen, q, tlen, t, score_in, P->w, M, X, I, D, E, T, mem, tmp); if(mem.bad) okk = false; }
		y = wtz_bcast_aln(y);
		okk = __shfl((int)okk, 0, 64) != 0;
		WTZ_PROF_ADD(10, ptw); WTZ_PROF_CNT(8, 1);
	}
	if(R.lds_runs) WTZ_PROF_CNT(6, 1);
	if(!okk) R.ok = false;
	R.y = y;
}

/* ---- A9 with the K-sw1 gaps run by the whole wave (hzm_aln.h:1247-1302).  Every lane follows the anchor loop with
 *      the same x; CIGAR bookkeeping and the run-by-run z-mer alignment stay on lane 0.  lds: >= 8 KB. ---- */
/* FULL = false is the form of the first launch: only the one / two-columns-per-lane register DP (and the empty problem) are compiled
 * in; a window with a problem that needs anything else sets *defer and is redone from scratch by a FULL launch. */
template<bool FULL = true>
WTZ_D wtz_aln_t wtz_align_window_wave(const wtz_readview &pb1, const wtz_readview &pb2, const wtz_win_t &win, const wtz_zhit_t *anchors,
		wtz_cigar_t &cigar, wtz_cigar_t &tmp, const wtz_params_t *P, wtz_pool_t *pool, int32_t *lds, unsigned long long *cells, bool *ok, bool *defer = NULL){
	const int lane = (int)(threadIdx.x & 63);
	const int32_t M = P->M, I = P->O, D = P->O, E = P->E;
	/* LDS slice: 128 target words (1 KB), then the 4-bit trace of the register DP with its run list at the top end (7 KB) */
	wtz_wave_lds_t L; L.tb = (uint64_t*)lds; L.Hs = lds + 256; L.Es = lds + 768; L.PM = 511; L.tw = 128; L.qw = (uint32_t*)(lds + WTZ_WINALIGN_LDS_BYTES / 4);
	uint8_t *ztr = (uint8_t*)(lds + 256); const int32_t ztr_bytes = WTZ_WINALIGN_LDS_BYTES - 1024;
	wtz_swmem_t mem; wtz_swmem_init(mem, pool);
	wtz_aln_t x, y; memset(&x, 0, sizeof x);
	wtz_cigw_t Wc; Wc.v = &cigar; Wc.tail = 0;
	*ok = true;
	for(uint32_t i = win.anchors[0]; i < win.anchors[1]; i++){
		const wtz_zhit_t p = anchors[i];
		const int32_t off1 = (int32_t)ZH_OFF1(p), off2 = (int32_t)ZH_OFF2(p);
		if(x.aln == 0){ x.tb = x.te = off1; x.qb = x.qe = off2; }
		if(off1 < x.te) continue;
		if(off2 < x.qe) continue;
		const int32_t qlen = off2 - x.qe, tlen = off1 - x.te;
		const unsigned long long pt0 = WTZ_PROF_T();
		wtz_fixres_t R;
		wtz_fixed_problem_wave<FULL, false>(qlen, pb2.sub(x.qe, 1), tlen, pb1.sub(x.te, 1), x.score, P, L, ztr, ztr_bytes, tmp, mem, pool, cells, 0, R);
		if(!FULL && R.defer){ *defer = true; return x; }
		if(!R.ok){ *ok = false; return x; }
		y = R.y;
		const uint32_t n_runs = R.n_runs; const uint32_t *runs = R.runs; const bool lds_runs = R.lds_runs;
		WTZ_PROF_ADD(0, pt0);
		const unsigned long long pt1 = WTZ_PROF_T();
		int32_t stop = 0;
		if(lane == 0){
			x.score = y.score;
			x.aln += y.aln; x.mat += y.mat; x.mis += y.mis; x.ins += y.ins; x.del += y.del;
			x.te += y.te; x.qe += y.qe;
			if(lds_runs){ for(uint32_t k = n_runs; k-- > 0;){ const uint32_t r = runs[k]; wtz_cigw_push(Wc, r & 0xFu, r >> 4); } }
			else { for(uint32_t k = 0; k < tmp.n; k++){ const uint32_t r = tmp.a[k]; wtz_cigw_push(Wc, r & 0xFu, r >> 4); } }
			if(x.te < off1){ x.del += off1 - x.te; x.aln += off1 - x.te; wtz_cigw_push(Wc, 2, (uint32_t)(off1 - x.te)); x.te = off1; }
			if(x.qe < off2){ x.ins += off2 - x.qe; x.aln += off2 - x.qe; wtz_cigw_push(Wc, 1, (uint32_t)(off2 - x.qe)); x.qe = off2; }
			WTZ_PROF_ADD(48, pt1); WTZ_PROF_CNT(50, 1000); WTZ_PROF_CNT(51, n_runs); WTZ_PROF_CNT(52, (qlen > 0 ? qlen : 0)); WTZ_PROF_CNT(53, (tlen > 0 ? tlen : 0));
			const unsigned long long pt2 = WTZ_PROF_T(); (void)pt2;
			const uint32_t len1 = ZH_LEN1(p), len2 = ZH_LEN2(p);
			const wtz_seq_packed z1 = pb1.sub(off1, 1), z2 = pb2.sub(off2, 1);
			/* one pass that writes its runs; a z-mer pair that turns out not to align (aln == 0) is rolled back: the writer's
			 * state is its open run plus the vector length */
			const uint32_t keep_tail = Wc.tail, keep_n = cigar.n;
			if(len1 <= 64 && len2 <= 64){
				wtz_seq_reg2 r1, r2;
				r1.w0 = wtz_pack32(z1, 0, (int32_t)len1); r1.w1 = wtz_pack32(z1, 32, (int32_t)len1);
				r2.w0 = wtz_pack32(z2, 0, (int32_t)len2); r2.w1 = wtz_pack32(z2, 32, (int32_t)len2);
				if(len1 == len2 && r1.w0 == r2.w0 && r1.w1 == r2.w1){
					/* the two expanded z-mers are the same string (no homopolymer-length difference): every run matches in full, the
					 * runs merge into one M of the whole length (hzm_aln.h:278-314 run by run gives exactly that) */
					memset(&y, 0, sizeof y);
					y.aln = y.mat = (int32_t)len1; y.te = y.qe = (int32_t)len1; y.score = (int32_t)len1 * M;
					wtz_cigw_push(Wc, 0, len1);
				} else
				y = wtz_align_zmer_w(r1, len1, r2, len2, M, I, D, E, &Wc);
			} else {
				y = wtz_align_zmer_w(z1, len1, z2, len2, M, I, D, E, &Wc);
			}
			if(y.aln == 0){ Wc.tail = keep_tail; cigar.n = keep_n; }
			if(y.aln == 0) stop = 1;
			else {
				x.score += y.score;
				x.aln += y.aln; x.mat += y.mat; x.mis += y.mis; x.ins += y.ins; x.del += y.del;
				x.te += y.te; x.qe += y.qe;
			}
			WTZ_PROF_ADD(49, pt2); WTZ_PROF_CNT(54, len1 + len2);
		}
		x = wtz_bcast_aln(x);
		stop = __shfl(stop, 0, 64);
		WTZ_PROF_ADD(1, pt1);
		if(stop) break;
	}
	if(lane == 0) wtz_cigw_finish(Wc);
	return x;
}

/* ---- K-sw2, ksw_global2 (ksw.c:503-586), one wavefront per problem: rows run over the target, lanes over the query band.
 *      Same cell recurrence as the extensions with MINUS_INF sentinels, the lh3 first-row / first-column initialisation,
 *      no early exit; the score is H(tlen-1, qlen-1) and the traceback starts from that cell.  CIGAR on lane 0. ---- */
template<typename SQ, typename ST>
WTZ_D int32_t wtz_global_wave(int32_t qlen, const SQ &query, int32_t tlen, const ST &target, int32_t M, int32_t X,
		int32_t o_del, int32_t e_del, int32_t o_ins, int32_t e_ins, int32_t w, const wtz_wave_lds_t &L, wtz_trace_t &tr, wtz_pool_t *pool,
		wtz_cigar_t &cig, bool *ok){
	const int lane = (int)(threadIdx.x & 63);
	const int32_t PM = L.PM, oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	*ok = true;
	if(lane == 0) cig.n = 0;
	const int32_t n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	const int32_t C0 = (n_col + 63) / 64, C = (C0 | 1), C4 = (C + 3) / 4;
	const uint32_t zrow = (uint32_t)C4 * 256u;
	if(!wtz_trace_prepare(tr, pool, zrow, tlen, false)){ *ok = false; return 0; }
	uint8_t **zchunk = tr.chunk; uint8_t *z = NULL;
	{   /* stage the query [0, qlen) as 2-bit codes */
		const int32_t nw = (qlen + 31) / 32 + 1;
		for(int32_t ww = lane; ww < nw; ww += 64){
			uint64_t v = 0; const int32_t b0 = ww * 32;
			for(int32_t k = 0; k < 32 && b0 + k < qlen; k++) v |= ((uint64_t)query.at(b0 + k)) << (2 * k);
			L.tb[ww] = v;
		}
	}
	int32_t begp = 0, endp = 0, i, h_lastrow = 0, end_last = 0;
	const int32_t CE = C * (-e_ins);
	for(i = 0; i < tlen; i++){
		if((i & 63) == 0){
			const uint32_t ci = (uint32_t)i >> 6;
			unsigned long long za = 0;
			if(ci < tr.n_chunk){ z = zchunk[ci]; }
			else {
				if(lane == 0){ uint8_t *p = (uint8_t*)wtz_pool_alloc(pool, (size_t)zrow * 64); zchunk[ci] = p; za = (unsigned long long)(uintptr_t)p; }
				za = __shfl(za, 0, 64);
				z = (uint8_t*)(uintptr_t)za;
				if(z == NULL){ *ok = false; return 0; }
				tr.n_chunk = ci + 1;
			}
		}
		const int32_t beg = i > w ? i - w : 0;
		const int32_t end = i + w + 1 < qlen ? i + w + 1 : qlen;
		const uint32_t tbase = target.at(i);
		const int32_t j0 = beg + lane * C;
		uint64_t qbits = 0;
		int32_t agg = -0x7F000000;
		{
			int32_t saved = 0;
			for(int32_t k = 0; k < C; k++){
				const int32_t j = j0 + k;
				if((k & 31) == 0){        /* the lane's next 32 query bases (bands wider than 64 x 32 columns have more than 32 columns per lane) */
					const int32_t jq = j0 + k;
					const int32_t jj = jq < qlen ? jq : (qlen > 0 ? qlen - 1 : 0);
					const int32_t ww = jj >> 5, sh = (jj & 31) * 2;
					const uint64_t w0 = L.tb[ww], w1 = L.tb[ww + 1];
					qbits = sh ? ((w0 >> sh) | (w1 << (64 - sh))) : w0;
				}
				if(j < end){
					int32_t pred;
					if(i == 0){ pred = (j == 0) ? 0 : ((j <= w) ? -(o_ins + e_ins * j) : WTZ_MINUS_INF); }       /* eh[] initialisation, ksw.c:519-523 */
					else if(k == 0){
						if(j - 1 >= begp && j - 1 < endp) pred = L.Hs[(j - 1) & PM];
						else pred = (j == 0) ? -(o_del + e_del * i) : WTZ_MINUS_INF;
					} else pred = saved;
					saved = (i > 0 && j >= begp && j < endp) ? L.Hs[j & PM] : WTZ_MINUS_INF;
					const uint32_t qb = (uint32_t)(qbits >> (2 * (k & 31))) & 3u;
					const int32_t m = pred + ((tbase == qb) ? M : X);
					L.Hs[j & PM] = m;
					const int32_t cand = m - oe_ins + (C - 1 - k) * (-e_ins);
					agg = agg > cand ? agg : cand;
				}
			}
		}
		int32_t f_in;
		{
			const int32_t g = agg - lane * CE;
			const int32_t pm = wtz_wave_max_scan_excl(g, -0x7F000000);
			const int32_t from_prev = (lane == 0) ? -0x7F000000 : pm + (lane - 1) * CE;
			const int32_t from_init = WTZ_MINUS_INF + lane * CE;
			f_in = from_prev > from_init ? from_prev : from_init;
		}
		int32_t h_last = 0;
		{
			int32_t f = f_in; uint32_t zword = 0;
			uint32_t *zr = (uint32_t*)(z + (size_t)(i & 63) * zrow) + lane;
			for(int32_t k = 0; k < C; k++){
				const int32_t j = j0 + k;
				if(j < end){
					const int32_t m = L.Hs[j & PM];
					int32_t e = (j >= begp && j < endp) ? L.Es[j & PM] : WTZ_MINUS_INF;
					uint32_t d; int32_t h;
					if(m >= e){ d = 0; h = m; } else { d = 1; h = e; }
					if(h < f){ d = 2; h = f; }
					h_last = h;
					int32_t t = m - oe_del; e -= e_del; if(e > t) d |= 1u << 2; else e = t;
					t = m - oe_ins; f -= e_ins; if(f > t) d |= 2u << 4; else f = t;
					L.Hs[j & PM] = h; L.Es[j & PM] = e;
					zword |= d << (8 * (k & 3));
				}
				if((k & 3) == 3 || k == C - 1){ zr[(size_t)(k >> 2) * 64] = zword; zword = 0; }
			}
		}
		if(end > beg){ const int32_t lastlane = (end - 1 - beg) / C; h_lastrow = __builtin_amdgcn_readlane(h_last, __builtin_amdgcn_readfirstlane(lastlane)); }
		begp = beg; endp = end; end_last = end;
	}
	const int32_t score = (end_last == qlen) ? h_lastrow : ((qlen <= w) ? -(o_ins + e_ins * qlen) : WTZ_MINUS_INF);
	__threadfence_block();     /* the trace was written by lanes of THIS wave: ordering inside the wave is enough (an agent-scope fence would write back the whole L2) */
	if(lane == 0){
		uint32_t which = 0;
		int32_t ii = tlen - 1, k = (ii + w + 1 < qlen ? ii + w + 1 : qlen) - 1;
		while(ii >= 0 && k >= 0){
			const int32_t col = k - (ii > w ? ii - w : 0);
			const int32_t ln = col / C, kk = col - ln * C;
			const uint8_t zv = zchunk[ii >> 6][(size_t)(ii & 63) * zrow + (size_t)(kk >> 2) * 256 + (size_t)ln * 4 + (kk & 3)];
			which = (zv >> (which << 1)) & 3;
			if(which == 0){ wtz_cigar_push(cig, 0, 1); --ii; --k; }
			else if(which == 1){ wtz_cigar_push(cig, 2, 1); --ii; }
			else { wtz_cigar_push(cig, 1, 1); --k; }
		}
		if(ii >= 0) wtz_cigar_push(cig, 2, (uint32_t)(ii + 1));
		if(k >= 0) wtz_cigar_push(cig, 1, (uint32_t)(k + 1));
		wtz_cigar_reverse(cig.a, cig.n);
	}
	return score;
}

/* one step of the ksw_global2 traceback on lane 0, without branches (see wtz_walk_step): rows run over the target (its base comes from
 * the 32-base words held in VGPRs), columns over the query (2-bit words in LDS); `which` 0 / 1 / 2 = from H / E (deletion, row up) /
 * F (insertion, column left) */
typedef struct { int32_t ii, k; uint32_t which, run_op, run_len, nr; int32_t mat, mis; } wtz_gwalk_t;
WTZ_D void wtz_gwalk_step(wtz_gwalk_t &g, uint32_t nib, const uint32_t *qb32, uint32_t tw_lo, uint32_t tw_hi, uint32_t *runs){
	const uint32_t h3 = nib & 3u, h = h3 > 2u ? 2u : h3;
	const uint32_t wh = ((h | (nib & 4u) | ((nib & 8u) << 2)) >> (g.which << 1)) & 3u;
	const uint32_t op = (0x18u >> (wh << 1)) & 3u;                 /* 0 -> M, 1 -> D (2), 2 -> I (1) */
	const uint32_t qv = (qb32[g.k >> 4] >> ((g.k & 15) * 2)) & 3u;
	const int ts = __builtin_amdgcn_readfirstlane((g.ii & 2047) >> 5);
	const uint32_t tlo = (uint32_t)__builtin_amdgcn_readlane((int)tw_lo, ts), thi = (uint32_t)__builtin_amdgcn_readlane((int)tw_hi, ts);
	const uint32_t tv = (((g.ii & 16) ? thi : tlo) >> ((g.ii & 15) * 2)) & 3u;
	const uint32_t is0 = wh == 0 ? 1u : 0u, is1 = wh == 1 ? 1u : 0u, is2 = wh == 2 ? 1u : 0u, eq = qv == tv ? 1u : 0u;
	g.mat += (int32_t)(is0 & eq); g.mis += (int32_t)(is0 & (eq ^ 1u));
	g.ii -= (int32_t)(is0 | is1); g.k -= (int32_t)(is0 | is2);
	const bool same = op == g.run_op;
	g.nr += (!same && g.run_len) ? 1u : 0u;
	g.run_len = same ? g.run_len + 1u : 1u; g.run_op = op; g.which = wh;
	runs[g.nr] = (g.run_len << 4) | op;
}

/*
 * K-sw2 (ksw_global2) for gaps whose band fits two columns per lane: the register form of wtz_global_wave.  Rows run over
 * the target (its row base is scalar, from 32-base words held in VGPRs), lanes over the query band (2-bit words in LDS);
 * the fixed band moves right by one column per row once i > w, so the hand-over is the same DPP move as in K-sw1; the
 * trace is 4 bits per cell in LDS.  Besides the CIGAR runs (traceback order, in `runs`) it returns the match / mismatch
 * counts of the M runs, which the caller would otherwise have to recount base by base (hzm_aln.h:1424-1436).
 * Requirements (caller): n_col <= 64*C, (qlen+63)/32+1 <= qb words; LDS trace (ZG false): ((tlen+1)/2)*zrow + 4*(qlen+tlen+4) <= LDS
 * trace bytes and tlen <= 2048; pool trace (ZG true): any tlen, `stage` = 4 KB of LDS.
 */
template<int C, bool ZG = false>
WTZ_D int32_t wtz_global_reg(int32_t qlen, const wtz_seq_packed &query, int32_t tlen, const wtz_seq_packed &target, int32_t M, int32_t X,
		int32_t o_del, int32_t e_del, int32_t o_ins, int32_t e_ins, int32_t w, uint64_t *qb, uint8_t *ztr, uint32_t zrow,
		uint32_t *runs, uint32_t *n_runs, int32_t *n_mat, int32_t *n_mis, uint8_t *stage = NULL){
	/* ZG: the trace does not fit LDS and lives in the pool (HBM); the rows store it through a global-address-space pointer and
	 * the traceback walks LDS-staged blocks of 32 packed rows (64 DP rows) x the whole band (<= 4 KB in `stage`) */
	const int lane = (int)(threadIdx.x & 63);
	const int32_t oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	*n_runs = 0; *n_mat = 0; *n_mis = 0;
	{
		const int32_t nw = (qlen + 31) / 32 + 1;
		for(int32_t ww = lane; ww < nw; ww += 64) qb[ww] = wtz_pack32(query, ww * 32, qlen);
	}
	__threadfence_block();
	const uint32_t *qb32 = (const uint32_t*)qb;
	int32_t hp[C], ep[C]; uint32_t nibp[C];
	#pragma unroll
	for(int k = 0; k < C; k++){ hp[k] = WTZ_MINUS_INF; ep[k] = WTZ_MINUS_INF; nibp[k] = 0; }
	const int32_t CE = C * (-e_ins);
	const int32_t colrel0 = lane * C;
	uint32_t tw_lo = 0, tw_hi = 0, tcur = 0, qwin = 0; int32_t qwin_left = 0;
	int32_t begp = 0, i, h_lastrow = 0, end_last = 0;
	for(i = 0; i < tlen; i++){
		const int32_t beg = i > w ? i - w : 0;
		const int32_t end = i + w + 1 < qlen ? i + w + 1 : qlen;
		if((i & 2047) == 0){ const uint64_t v = wtz_pack32(target, i + lane * 32, tlen); tw_lo = (uint32_t)v; tw_hi = (uint32_t)(v >> 32); }
		if((i & 15) == 0){
			const int32_t ts = __builtin_amdgcn_readfirstlane((i & 2047) >> 5);
			tcur = (i & 16) ? (uint32_t)__builtin_amdgcn_readlane((int)tw_hi, ts) : (uint32_t)__builtin_amdgcn_readlane((int)tw_lo, ts);
		}
		const uint32_t tbase = (tcur >> ((i & 15) * 2)) & 3u;
		const int32_t j0 = beg + colrel0;
		const bool moved = (i > 0) && (beg != begp);
		if(moved){ qwin >>= 2; qwin_left--; }
		if(i == 0 || qwin_left < C){
			const int32_t jj = j0 < qlen ? j0 : (qlen > 0 ? qlen - 1 : 0);
			const uint32_t lo = qb32[jj >> 4], hi = qb32[(jj >> 4) + 1];
			qwin = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(jj & 15) * 2u);
			qwin_left = 16;
		}
		int32_t pred[C], ein[C];
		if(i == 0){
			#pragma unroll
			for(int k = 0; k < C; k++){ const int32_t j = j0 + k; pred[k] = (j == 0) ? 0 : ((j <= w) ? -(o_ins + e_ins * j) : WTZ_MINUS_INF); ein[k] = WTZ_MINUS_INF; }      /* eh[] initialisation, ksw.c:519-523 */
		} else if(moved){
			const int32_t nxt = wtz_dpp_wave_shl1(WTZ_MINUS_INF, ep[0]);
			#pragma unroll
			for(int k = 0; k < C; k++){ pred[k] = hp[k]; ein[k] = (k + 1 < C) ? ep[k + 1] : nxt; }
		} else {
			int32_t prv = wtz_dpp_wave_shr1(WTZ_MINUS_INF, hp[C - 1]);
			prv = (lane == 0) ? -(o_del + e_del * i) : prv;             /* first column, ksw.c:533 */
			#pragma unroll
			for(int k = 0; k < C; k++){ pred[k] = k ? hp[k - 1] : prv; ein[k] = ep[k]; }
		}
		int32_t mv[C]; bool valid[C]; int32_t agg = -0x7F000000;
		#pragma unroll
		for(int k = 0; k < C; k++){
			const uint32_t qbase = (qwin >> (2 * k)) & 3u;
			mv[k] = pred[k] + ((tbase == qbase) ? M : X);
			valid[k] = (j0 + k < end);
			const int32_t cand = mv[k] - oe_ins + (C - 1 - k) * (-e_ins);
			agg = (valid[k] && cand > agg) ? cand : agg;
		}
		int32_t f;
		{
			const int32_t g = agg - lane * CE;
			const int32_t pm = wtz_wave_max_scan_excl(g, -0x7F000000);
			const int32_t from_prev = (lane == 0) ? -0x7F000000 : pm + (lane - 1) * CE;
			const int32_t from_init = WTZ_MINUS_INF + lane * CE;
			f = from_prev > from_init ? from_prev : from_init;
		}
		#pragma unroll
		for(int k = 0; k < C; k++){
			const int32_t m = mv[k], e = ein[k];
			const int32_t h0 = m > e ? m : e;
			const int32_t te = m - oe_del, e2 = e - e_del, tf = m - oe_ins, f2 = f - e_ins;
			/* decisions as sign bits of differences (all values within +-2^30 + a few thousand: no overflow), see wtz_extend_fixed_reg */
			uint32_t nib = (uint32_t)(tf - f2) >> 31;
			nib = __builtin_amdgcn_alignbit(nib, (uint32_t)(te - e2), 31);
			nib = __builtin_amdgcn_alignbit(nib, (uint32_t)(h0 - f), 31);
			nib = __builtin_amdgcn_alignbit(nib, (uint32_t)(m - e), 31);
			const int32_t h = h0 > f ? h0 : f;
			const int32_t en = e2 > te ? e2 : te;
			f = f2 > tf ? f2 : tf;
			hp[k] = valid[k] ? h : WTZ_MINUS_INF; ep[k] = valid[k] ? en : WTZ_MINUS_INF;
			nib = valid[k] ? nib : 0u;
			if(i & 1){
				if((uint32_t)(colrel0 + k) < zrow){
					if(ZG) wtz_as_global(ztr)[(size_t)(i >> 1) * zrow + colrel0 + k] = (uint8_t)(nibp[k] | (nib << 4));
					else ztr[(size_t)(i >> 1) * zrow + colrel0 + k] = (uint8_t)(nibp[k] | (nib << 4));
				}
			}
			else nibp[k] = nib;
		}
		if(end > beg && i + 1 == tlen){
			const int32_t idx = end - 1 - beg;
			int32_t hsel = hp[0];
			#pragma unroll
			for(int k = 1; k < C; k++) hsel = (idx % C == k) ? hp[k] : hsel;
			h_lastrow = __builtin_amdgcn_readlane(hsel, __builtin_amdgcn_readfirstlane(idx / C));
		}
		begp = beg; end_last = end;
	}
	if(tlen > 0 && !((tlen - 1) & 1)){
		#pragma unroll
		for(int k = 0; k < C; k++) if((uint32_t)(colrel0 + k) < zrow) ztr[(size_t)((tlen - 1) >> 1) * zrow + colrel0 + k] = (uint8_t)nibp[k];
	}
	const int32_t score = (end_last == qlen) ? h_lastrow : ((qlen <= w) ? -(o_ins + e_ins * qlen) : WTZ_MINUS_INF);
	__threadfence_block();
	if(ZG){
		wtz_gwalk_t g; g.which = 0; g.nr = 0; g.run_op = 0xFFu; g.run_len = 0; g.mat = g.mis = 0;
		g.ii = tlen - 1; g.k = (g.ii + w + 1 < qlen ? g.ii + w + 1 : qlen) - 1;
		int32_t ii = g.ii, k = g.k;
		uint32_t *stage32 = (uint32_t*)stage;
		const int32_t RB = zrow <= 128 ? 32 : (zrow <= 256 ? 16 : 8);         /* packed rows per staged block: RB * zrow <= 4 KB */
		int32_t cur_tblk = (tlen - 1) >> 11;                                    /* the 2048-row block of target words the DP loop left in tw_lo / tw_hi */
		while(ii >= 0 && k >= 0){
			const int32_t p1 = ii >> 1, p0 = p1 >= RB - 1 ? p1 - (RB - 1) : 0;
			if((ii >> 11) != cur_tblk){
				cur_tblk = ii >> 11;
				const uint64_t v = wtz_pack32(target, (cur_tblk << 11) + lane * 32, tlen); tw_lo = (uint32_t)v; tw_hi = (uint32_t)(v >> 32);
			}
			{
				const uint32_t nd = (uint32_t)(p1 - p0 + 1) * (zrow >> 2);
				const uint32_t *src = (const uint32_t*)(ztr + (size_t)p0 * zrow);
				for(uint32_t xw = (uint32_t)lane; xw < nd; xw += 64) stage32[xw] = src[xw];
			}
			__threadfence_block();
			if(lane == 0){
				while((g.ii | g.k) >= 0 && (g.ii >> 1) >= p0 && (g.ii >> 11) == cur_tblk){
					const int32_t col = g.k - (g.ii > w ? g.ii - w : 0);
					const uint32_t zv = stage[(size_t)((g.ii >> 1) - p0) * zrow + col];
					wtz_gwalk_step(g, zv >> ((g.ii & 1) * 4), (const uint32_t*)qb, tw_lo, tw_hi, runs);
				}
			}
			ii = __builtin_amdgcn_readfirstlane(g.ii); k = __builtin_amdgcn_readfirstlane(g.k);
			__threadfence_block();
		}
		if(lane == 0){
			uint32_t run_op = g.run_op, run_len = g.run_len, nr = g.nr;
			if(ii >= 0){ if(run_len && run_op == 2u) run_len += (uint32_t)(ii + 1); else { if(run_len) runs[nr++] = (run_len << 4) | run_op; run_op = 2u; run_len = (uint32_t)(ii + 1); } }
			if(k >= 0){ if(run_len && run_op == 1u) run_len += (uint32_t)(k + 1); else { if(run_len) runs[nr++] = (run_len << 4) | run_op; run_op = 1u; run_len = (uint32_t)(k + 1); } }
			if(run_len) runs[nr++] = (run_len << 4) | run_op;
			*n_runs = nr; *n_mat = g.mat; *n_mis = g.mis;
		}
		return score;
	}
	if(lane == 0){
		wtz_gwalk_t g; g.which = 0; g.nr = 0; g.run_op = 0xFFu; g.run_len = 0; g.mat = g.mis = 0;
		g.ii = tlen - 1; g.k = (g.ii + w + 1 < qlen ? g.ii + w + 1 : qlen) - 1;
		while((g.ii | g.k) >= 0){
			const int32_t col = g.k - (g.ii > w ? g.ii - w : 0);
			const uint32_t zv = ztr[(size_t)(g.ii >> 1) * zrow + col];
			wtz_gwalk_step(g, zv >> ((g.ii & 1) * 4), (const uint32_t*)qb, tw_lo, tw_hi, runs);
		}
		const int32_t ii = g.ii, k = g.k; const int32_t mat = g.mat, mis = g.mis;
		uint32_t run_op = g.run_op, run_len = g.run_len, nr = g.nr;
		if(ii >= 0){ if(run_len && run_op == 2u) run_len += (uint32_t)(ii + 1); else { if(run_len) runs[nr++] = (run_len << 4) | run_op; run_op = 2u; run_len = (uint32_t)(ii + 1); } }
		if(k >= 0){ if(run_len && run_op == 1u) run_len += (uint32_t)(k + 1); else { if(run_len) runs[nr++] = (run_len << 4) | run_op; run_op = 1u; run_len = (uint32_t)(k + 1); } }
		if(run_len) runs[nr++] = (run_len << 4) | run_op;
		*n_runs = nr; *n_mat = mat; *n_mis = mis;
	}
	return score;
}

/*
 * A11, the optional `-n` pass: kswx_refine_alignment (kswx.h:483-659) on one wavefront.
 *   band   lane 0 derives the per-row half-width zw[] and the band [zb, ze) from the stitched CIGAR exactly as the reference
 *          does (the `zw[qx] += len` of a deletion is overwritten by the next M/I row and has no effect), then the running
 *          max / min trims; rows keep a byte offset into a band-relative trace (the reference's ql x tl matrix is only ever
 *          touched inside the band);
 *   rows   the wave DP of wtz_extend_wave (LDS H/E rings indexed by absolute column, DPP max-scan for F) with the global
 *          boundary: H(-1,-1) = 0, everything else outside the previous row's band is -10000 - which is what the
 *          reference's never re-initialised rh[] / re[] rows hold, because zb and ze are non-decreasing;
 *   trace  lane 0 walks from (ql-1, tl-1) through LDS-staged blocks of 64 rows x 124 columns.
 * query = candidate view from qb, target = query-read view from tb (wtzmo.c:1033).  Returns false when the pool ran dry.
 * Fallback inside: none needed for band widths up to 64*31 columns; wider bands (an indel of > 900 bases) run the DP rows on
 * lane 0 with the same rings.
 */
WTZ_D bool wtz_refine_wave(const wtz_seq_packed &query, int32_t qb, const wtz_seq_packed &target, int32_t tb, int32_t W, int32_t M, int32_t X, int32_t I, int32_t D, int32_t E,
		const uint32_t *cig, uint32_t ncig, const wtz_wave_lds_t &L, wtz_pool_t *pool, wtz_cigar_t &out, wtz_aln_t *res){
	const int lane = (int)(threadIdx.x & 63);
	const int32_t PM = L.PM;
	wtz_aln_t y; memset(&y, 0, sizeof y);
	if(lane == 0) out.n = 0;
	const unsigned long long pr0 = WTZ_PROF_T(); (void)pr0;
	int32_t qe = qb, te = tb;
	for(uint32_t i = 0; i < ncig; i++){ const uint32_t op = cig[i] & 0xFu; const int32_t len = (int32_t)(cig[i] >> 4); if(op == 0){ qe += len; te += len; } else if(op == 1) qe += len; else te += len; }
	const int32_t ql = qe - qb, tl = te - tb;
	if(ql == 0 || tl == 0){ *res = y; return true; }        /* KSWX_NULL, empty CIGAR */
	WTZ_PROF_ADD(24, pr0);
	const unsigned long long pr1 = WTZ_PROF_T(); (void)pr1;
	/* ---- band ---- */
	unsigned long long ba = 0;
	if(lane == 0) ba = (unsigned long long)(uintptr_t)wtz_pool_alloc(pool, (size_t)(ql + 2) * (4 * 3 + 8));
	ba = __shfl(ba, 0, 64);
	if(ba == 0) return false;
	/* the 64-bit array leads the (16-byte aligned) block so that it is 8-byte aligned for every ql */
	unsigned long long *zoff = (unsigned long long*)(uintptr_t)ba;
	int32_t *zw = (int32_t*)(zoff + (ql + 2));
	int32_t *zb = zw + (ql + 2), *ze = zb + (ql + 2);
	for(int32_t i = lane; i < ql + 2; i += 64) zw[i] = 0;
	__threadfence_block();
	unsigned long long ztot = 0; int32_t maxw = 0;
	{
		/* every pass of the reference over the CIGAR (kswx.h:536-611) is data-parallel once each operation knows its first row /
		 * first column: one exclusive scan; rows are then written per operation, the widening sums are order-free (atomicAdd),
		 * and the two trims are a running maximum and a suffix minimum */
		unsigned long long pa = 0;
		if(lane == 0) pa = (unsigned long long)(uintptr_t)wtz_pool_alloc(pool, (size_t)(ncig + 2) * 8);
		pa = __shfl(pa, 0, 64);
		uint32_t *qx0 = (uint32_t*)(uintptr_t)pa;
		if(qx0 == NULL) return false;
		uint32_t *tx0 = qx0 + (ncig + 2);
		{
			uint32_t cq = 0, ct = 0;
			for(uint32_t i0 = 0; i0 < ncig; i0 += 64){
				const uint32_t i = i0 + lane;
				uint32_t dq = 0, dt = 0;
				if(i < ncig){ const uint32_t op = cig[i] & 0xFu, len = cig[i] >> 4; dq = (op == 0 || op == 1) ? len : 0; dt = (op == 0 || op == 2) ? len : 0; }
				uint32_t tq, tt; const uint32_t eq = wtz_coop_excl_scan(dq, &tq), et = wtz_coop_excl_scan(dt, &tt);
				if(i < ncig){ qx0[i] = cq + eq; tx0[i] = ct + et; }
				cq += tq; ct += tt;
			}
		}
		__threadfence_block();
		for(uint32_t i0 = 0; i0 < ncig; i0 += 64){        /* basic half-width: W on M rows, W + len on the rows of an insertion */
			const uint32_t i = i0 + lane;
			if(i < ncig){
				const uint32_t op = cig[i] & 0xFu; const int32_t len = (int32_t)(cig[i] >> 4);
				if(op == 0){ for(int32_t j = 0; j < len; j++) zw[qx0[i] + j] = W; }
				else if(op == 1){ for(int32_t j = 0; j < len; j++) zw[qx0[i] + j] = W + len; }
			}
		}
		__threadfence_block();
		for(uint32_t i0 = 0; i0 < ncig; i0 += 64){        /* widening around indels */
			const uint32_t i = i0 + lane;
			if(i < ncig){
				const uint32_t op = cig[i] & 0xFu; const int32_t len = (int32_t)(cig[i] >> 4);
				int32_t qx = (int32_t)qx0[i];
				if(op == 1){
					for(int32_t j = 1; j < len && j < qx; j++) atomicAdd(&zw[qx - j], len - j);
					qx += len - 1;
					for(int32_t j = 1; j < len && j + qx < ql; j++) atomicAdd(&zw[qx + j], len - j);
				} else if(op == 2){
					for(int32_t j = 1; j < len && j < qx; j++) atomicAdd(&zw[qx - j], len - j);
					for(int32_t j = 1; j < len && j + qx < ql; j++) atomicAdd(&zw[qx + j], len - j);
				}
			}
		}
		__threadfence_block();
		for(uint32_t i0 = 0; i0 < ncig; i0 += 64){        /* band of every row */
			const uint32_t i = i0 + lane;
			if(i < ncig){
				const uint32_t op = cig[i] & 0xFu; const int32_t len = (int32_t)(cig[i] >> 4);
				if(op == 0 || op == 1){
					for(int32_t j = 0; j < len; j++){
						const int32_t row = (int32_t)qx0[i] + j, tx = (int32_t)tx0[i] + (op == 0 ? j : 0);
						const int32_t hw = __hip_atomic_load(&zw[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      /* the sums were made at L2 */
						int32_t b = tx - hw; if(b < 0) b = 0;
						int32_t e = tx + 1 + hw; if(e > tl) e = tl;
						zb[row] = b; ze[row] = e;
					}
				}
			}
		}
		__threadfence_block();
		{   /* zb: running maximum (starting from 0) */
			int32_t carry = 0;
			for(int32_t r0 = 0; r0 < ql; r0 += 64){
				const int32_t r = r0 + lane;
				const int32_t v = r < ql ? zb[r] : -0x7FFFFFFF;
				int32_t ex = wtz_wave_max_scan_excl(v, -0x7FFFFFFF);
				int32_t inc = ex > v ? ex : v; inc = inc > carry ? inc : carry;
				if(r < ql) zb[r] = inc;
				carry = __builtin_amdgcn_readlane(inc, 63);
			}
		}
		{   /* ze: suffix minimum (starting from tl), as a running maximum of the negated values from the end */
			int32_t carry = -tl;
			for(int32_t r0 = 0; r0 < ql; r0 += 64){
				const int32_t r = ql - 1 - (r0 + lane);
				const int32_t v = r >= 0 ? -ze[r] : -0x7FFFFFFF;
				int32_t ex = wtz_wave_max_scan_excl(v, -0x7FFFFFFF);
				int32_t inc = ex > v ? ex : v; inc = inc > carry ? inc : carry;
				if(r >= 0) ze[r] = -inc;
				carry = __builtin_amdgcn_readlane(inc, 63);
			}
		}
		__threadfence_block();
		{   /* trace row offsets and the widest row */
			uint32_t carry = 0;
			for(int32_t r0 = 0; r0 < ql; r0 += 64){
				const int32_t r = r0 + lane;
				const uint32_t w = (r < ql && ze[r] > zb[r]) ? (uint32_t)(ze[r] - zb[r]) : 0u;
				uint32_t tot; const uint32_t ex = wtz_coop_excl_scan(w, &tot);
				if(r < ql) zoff[r] = (unsigned long long)carry + ex;
				carry += tot;
				int32_t mw = (int32_t)w; mw = wtz_wave_max_i32(mw); maxw = mw > maxw ? mw : maxw;
			}
			ztot = carry;
		}
	}
	unsigned long long za = 0;
	if(lane == 0) za = (unsigned long long)(uintptr_t)wtz_pool_alloc(pool, (size_t)ztot + 64);
	za = __shfl(za, 0, 64);
	uint8_t *z = (uint8_t*)(uintptr_t)za;
	if(z == NULL) return false;
	__threadfence_block();
	WTZ_PROF_ADD(25, pr1);
	const unsigned long long pr2 = WTZ_PROF_T(); (void)pr2;
	/* ---- rows ---- */
	const int32_t C = (((maxw + 63) / 64) | 1);
	const bool wave_rows = (maxw + 2 <= PM + 1) && C <= 31 && (tl + 63) / 32 + 1 <= L.tw;
	int32_t h_last_row = -10000;
	if(wave_rows){
		const int32_t nw = (tl + 31) / 32 + 1;
		for(int32_t w = lane; w < nw; w += 64) L.tb[w] = wtz_pack32(target, w * 32, tl);
		__threadfence_block();
		int32_t jbp = 0, jep = 0; uint32_t qw_lo = 0, qw_hi = 0, qcur = 0;
		int32_t rb = 0, re_ = 0; uint32_t ro_lo = 0, ro_hi = 0;
		const int32_t CE = C * E;
		for(int32_t i = 0; i < ql; i++){
			if((i & 2047) == 0){ const uint64_t qw = wtz_pack32(query, i + lane * 32, ql); qw_lo = (uint32_t)qw; qw_hi = (uint32_t)(qw >> 32); }
			if((i & 15) == 0){ const int32_t qs = __builtin_amdgcn_readfirstlane((i & 2047) >> 5); qcur = (i & 16) ? (uint32_t)__builtin_amdgcn_readlane((int)qw_hi, qs) : (uint32_t)__builtin_amdgcn_readlane((int)qw_lo, qs); }
			const uint32_t qbase = (qcur >> ((i & 15) * 2)) & 3u;
			if((i & 63) == 0){      /* band and trace offset of the next 64 rows: one coalesced load each, then v_readlane per row */
				const int32_t r = i + lane;
				rb = r < ql ? zb[r] : 0; re_ = r < ql ? ze[r] : 0;
				const unsigned long long zo = r < ql ? zoff[r] : 0ull; ro_lo = (uint32_t)zo; ro_hi = (uint32_t)(zo >> 32);
			}
			const int32_t rl = __builtin_amdgcn_readfirstlane(i & 63);
			const int32_t jb = __builtin_amdgcn_readlane(rb, rl), je = __builtin_amdgcn_readlane(re_, rl);
			const unsigned long long zrow_off = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)ro_hi, rl) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)ro_lo, rl);
			const int32_t j0 = jb + lane * C;
			uint64_t tbits;
			{
				const int32_t jj = j0 < tl ? j0 : (tl > 0 ? tl - 1 : 0);
				const int32_t w = jj >> 5, sh = (jj & 31) * 2;
				const uint64_t w0 = L.tb[w], w1 = L.tb[w + 1];
				tbits = sh ? ((w0 >> sh) | (w1 << (64 - sh))) : w0;
			}
			int32_t agg = -0x3FFFFFFF;
			{
				int32_t saved = 0;
				for(int32_t k = 0; k < C; k++){
					const int32_t j = j0 + k;
					if(j < je){
						int32_t pred;
						if(k == 0){
							if(i == 0) pred = (j == 0) ? 0 : -10000;                                  /* rh[0] = 0, rh[1..] = -10000 (kswx.h:613-615) */
							else pred = (j - 1 >= jbp && j - 1 < jep) ? L.Hs[(j - 1) & PM] : -10000;
						} else pred = (i == 0) ? -10000 : saved;
						saved = (i > 0 && j >= jbp && j < jep) ? L.Hs[j & PM] : -10000;
						const uint32_t tbase = (uint32_t)(tbits >> (2 * k)) & 3u;
						const int32_t m = pred + ((qbase == tbase) ? M : X);
						L.Hs[j & PM] = m;
						const int32_t cand = m + D + E + (C - 1 - k) * E;
						agg = agg > cand ? agg : cand;
					}
				}
			}
			int32_t f;
			{
				const int32_t g = agg - lane * CE;
				const int32_t pm = wtz_wave_max_scan_excl(g, -0x3FFFFFFF);
				const int32_t from_prev = (lane == 0) ? -0x3FFFFFFF : pm + (lane - 1) * CE;
				const int32_t from_init = -10000 + lane * CE;
				f = from_prev > from_init ? from_prev : from_init;
			}
			int32_t h_last = 0;
			uint8_t *zi = z + zrow_off;
			for(int32_t k = 0; k < C; k++){
				const int32_t j = j0 + k;
				if(j < je){
					const int32_t m = L.Hs[j & PM];
					int32_t e = (i > 0 && j >= jbp && j < jep) ? L.Es[j & PM] : -10000;
					uint32_t d; int32_t h;
					if(m >= e){ d = 0; h = m; } else { d = 1; h = e; }
					if(h < f){ d = 2; h = f; }
					h_last = h;
					int32_t t = m + I + E; e = e + E; if(e > t) d |= 1u << 2; else e = t;
					t = m + D + E; f = f + E; if(f > t) d |= 2u << 4; else f = t;
					if(qbase == ((uint32_t)(tbits >> (2 * k)) & 3u)) d |= 0x80u;
					L.Hs[j & PM] = h; L.Es[j & PM] = e;
					zi[j - jb] = (uint8_t)d;
				}
			}
			if(je > jb && i + 1 == ql){ const int32_t lastlane = (je - 1 - jb) / C; h_last_row = __builtin_amdgcn_readlane(h_last, __builtin_amdgcn_readfirstlane(lastlane)); }
			jbp = jb; jep = je;
		}
	} else {
		/* very wide band: the same rows on lane 0, straight from the reference's loop with pool-resident rh / re */
		unsigned long long ra = 0;
		if(lane == 0) ra = (unsigned long long)(uintptr_t)wtz_pool_alloc(pool, (size_t)(tl + 2) * 8);
		ra = __shfl(ra, 0, 64);
		int32_t *rh = (int32_t*)(uintptr_t)ra;
		if(rh == NULL) return false;
		if(lane == 0){
			int32_t *re = rh + (tl + 2);
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
			h_last_row = rh[tl];
		}
		h_last_row = __shfl(h_last_row, 0, 64);
	}
	y.qb = qb; y.qe = qe; y.tb = tb; y.te = te;
	y.score = wave_rows ? ((ze[ql - 1] == tl && ze[ql - 1] > zb[ql - 1]) ? h_last_row : -10000) : h_last_row;       /* rh[tl] */
	__threadfence_block();
	WTZ_PROF_ADD(26, pr2); WTZ_PROF_CNT(29, ql); WTZ_PROF_CNT(30, 1);
	const unsigned long long pr3 = WTZ_PROF_T(); (void)pr3;
	/* ---- traceback through LDS-staged blocks (the rings / target words are dead) ---- */
	{
		constexpr int WC = 124;
		uint8_t *S = (uint8_t*)L.tb;
		int32_t i_ = ql - 1, j_ = tl - 1; uint32_t d_ = 0;
		uint32_t run_op = 0xFFu, run_len = 0;
		wtz_cigw_t Wr; Wr.v = &out; Wr.tail = 0;
		while(i_ >= 0 && j_ >= 0){
			const int32_t i0 = i_, jlo = j_ - (WC - 1);
			{
				const int32_t r = i0 - lane;
				if(r >= 0){
					const int32_t b = zb[r], e = ze[r]; const uint8_t *rowp = z + zoff[r];
					for(int32_t t = 0; t < WC; t++){ const int32_t jj = jlo + t; S[lane * WC + t] = (jj >= b && jj < e) ? rowp[jj - b] : (uint8_t)0; }
				}
			}
			__threadfence_block();
			if(lane == 0){
				while(i_ >= 0 && j_ >= 0){
					const int32_t rr = i0 - i_, t = j_ - jlo;
					if(rr >= 64 || t < 0) break;
					const uint32_t zv = S[rr * WC + t];
					d_ = (zv >> (d_ << 1)) & 0x03;
					if(d_ == 0){ if(zv & 0x80u) y.mat++; else y.mis++; i_--; j_--; }
					else if(d_ == 1){ i_--; y.ins++; }
					else { j_--; y.del++; }
					if(d_ == run_op) run_len++;
					else { if(run_len) wtz_cigw_push(Wr, run_op, run_len); run_op = d_; run_len = 1; }
				}
			}
			i_ = __builtin_amdgcn_readfirstlane(i_); j_ = __builtin_amdgcn_readfirstlane(j_);
			__threadfence_block();
		}
		if(lane == 0){
			if(run_len) wtz_cigw_push(Wr, run_op, run_len);
			if(i_ >= 0){ y.ins += i_ + 1; wtz_cigw_push(Wr, 1, (uint32_t)(i_ + 1)); }
			if(j_ >= 0){ y.del += j_ + 1; wtz_cigw_push(Wr, 2, (uint32_t)(j_ + 1)); }
			wtz_cigw_finish(Wr);
			wtz_cigar_reverse(out.a, out.n);
			y.aln = y.mat + y.mis + y.ins + y.del;
		}
	}
	WTZ_PROF_ADD(27, pr3);
	*res = wtz_bcast_aln(y);
	return true;
}

#endif /* __HIPCC__ */
#endif
