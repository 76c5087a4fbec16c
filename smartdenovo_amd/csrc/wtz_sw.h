/*
 * wtz_sw.h — banded dynamic programs of the zmo engine, scalar (one lane per problem) form.
 *
 *   wtz_extend_fixed   K-sw1  kswx.h:234-335  kswx_extend_align_core        (fixed band)
 *   wtz_extend_shift   K-sw3  kswx.h:101-232  kswx_extend_align_shift_core  (band follows the row arg-max)
 *   wtz_global_banded  K-sw2  ksw.c:503-586   ksw_global2
 *   wtz_align_zmer            hzm_aln.h:278-314 hz_align_hzmo
 *   wtz_align_window   A9     hzm_aln.h:1247-1302 fast_seeds_align_hzmo
 *
 * Recurrence (int32, -10000 / -0x40000000 sentinels are part of the semantics):
 *   m = H(i-1,j-1)+S;  H = max{m,E,F} (ties m>E>F);  E' = max{E+e, m+I+e};  F' = max{F+e, m+D+e}
 * one trace byte per cell: bits0-1 H source, bit2 E extended, bit5 F extended.
 * The wave-parallel kernels in wtz_sw_wave.h compute the same cells row by row with lanes
 * across the band and a max-plus prefix scan for F; these scalar bodies are their on-device
 * cross-check and the form used for the many tiny K-sw1/K-sw2 problems.
 */
#ifndef WTZ_SW_H
#define WTZ_SW_H

#include "wtz_window.h"

typedef wtz_vec<uint32_t> wtz_cigar_t;

WTZ_HD void wtz_cigar_push(wtz_cigar_t &c, uint32_t op, uint32_t len){          /* kswx.h:39-44 */
	if(len == 0) return;
	if(c.n && (c.a[c.n - 1] & 0xF) == op) c.a[c.n - 1] += len << 4;
	else c.push((len << 4) | op);
}
WTZ_HD void wtz_cigar_concat(wtz_cigar_t &c, const uint32_t *src, uint32_t n){  /* kswx.h:46-52 */
	if(n == 0) return;
	uint32_t k = 0;
	if(c.n && (c.a[c.n - 1] & 0xF) == (src[0] & 0xFu)){ c.a[c.n - 1] += src[0] & 0xFFFFFFF0u; k = 1; }
	if(!c.reserve(c.n + n)) return;
	for(; k < n; k++) c.a[c.n++] = src[k];
}
WTZ_HD void wtz_cigar_reverse(uint32_t *a, uint32_t n){ for(uint32_t i = 0; i < n / 2; i++){ uint32_t t = a[i]; a[i] = a[n - 1 - i]; a[n - 1 - i] = t; } }

/* scratch for one DP: row arrays + trace matrix, (re)carved from the pool on demand */
typedef struct { int32_t *rh, *re, *rm, *zb; uint8_t *z; uint32_t cap_row, cap_zb; uint64_t cap_z; wtz_pool_t *pool; int bad; } wtz_swmem_t;
WTZ_HD void wtz_swmem_init(wtz_swmem_t &m, wtz_pool_t *pool){ m.rh = m.re = m.rm = m.zb = NULL; m.z = NULL; m.cap_row = m.cap_zb = 0; m.cap_z = 0; m.pool = pool; m.bad = 0; }
/* row arrays of the scalar DP bodies in the wave's LDS slice when they fit (flat addressing): the H/E rows are
 * touched twice per cell by a single lane, so their latency, not bandwidth, bounds K-sw1 / K-sw2 */
WTZ_HD void wtz_swmem_init_lds(wtz_swmem_t &m, wtz_pool_t *pool, int32_t *lds, uint32_t lds_ints){
	wtz_swmem_init(m, pool);
	if(lds && lds_ints >= 192){ m.rh = lds; m.re = lds + lds_ints / 3; m.rm = lds + 2 * (lds_ints / 3); m.cap_row = lds_ints / 3; }
}
WTZ_HD bool wtz_swmem_need(wtz_swmem_t &m, uint32_t row, uint32_t zb, uint64_t z){
	/* a failed request is sticky: the capacities only grow after BOTH pointers of a request exist, so a later problem of the same
	 * task can never run on a NULL row buffer after the pool ran dry (the task reports `bad`, the stage WTZ_E_POOL) */
	if(m.bad) return false;
	if(row > m.cap_row){ uint32_t c = m.cap_row ? m.cap_row : 64; while(c < row) c <<= 1;
		int32_t *rh = (int32_t*)wtz_pool_alloc(m.pool, (size_t)c * 4), *re = (int32_t*)wtz_pool_alloc(m.pool, (size_t)c * 4), *rm = (int32_t*)wtz_pool_alloc(m.pool, (size_t)c * 4);
		if(!rh || !re || !rm){ m.bad = 1; return false; }
		m.rh = rh; m.re = re; m.rm = rm; m.cap_row = c; }
	if(zb > m.cap_zb){ uint32_t c = m.cap_zb ? m.cap_zb : 64; while(c < zb) c <<= 1; int32_t *p = (int32_t*)wtz_pool_alloc(m.pool, (size_t)c * 4); if(!p){ m.bad = 1; return false; } m.zb = p; m.cap_zb = c; }
	if(z > m.cap_z){ uint64_t c = m.cap_z ? m.cap_z : 1024; while(c < z) c <<= 1; uint8_t *p = (uint8_t*)wtz_pool_alloc(m.pool, (size_t)c); if(!p){ m.bad = 1; return false; } m.z = p; m.cap_z = c; }
	return true;
}

WTZ_HD void wtz_ext_geometry(int32_t qlen, int32_t tlen, int32_t init_score, int32_t &W, int32_t M, int32_t I, int32_t D, int32_t E, int32_t T, int32_t &ql, int32_t &tl, int32_t &n_col){
	int32_t w = W;
	if(w > 0){
		int32_t mx = ((qlen < tlen) ? qlen : tlen) * M + init_score + (-T);
		int32_t max_gap = (mx + ((I > D) ? I : D)) / (-E) + 1;
		if(max_gap < 1) max_gap = 1;
		if(w > max_gap) w = max_gap;
	} else w = -w;
	w = WTZ_MIN(w, WTZ_MAX(qlen, tlen));
	if(qlen < tlen){ if(qlen + w < tlen){ ql = qlen; tl = qlen + w; } else { ql = qlen; tl = tlen; } }
	else           { if(tlen + w < qlen){ tl = tlen; ql = tlen + w; } else { tl = tlen; ql = qlen; } }
	n_col = (tl < 2 * w + 1) ? tl : 2 * w + 1;
	W = w;
}

/* sequence accessors: SEQ::at(p, i) returns the 2-bit base at logical index i */
struct wtz_seq_bytes { const uint8_t *p; int32_t strand; WTZ_HDM uint32_t at(int32_t i) const { return p[i * strand]; } };
/* 2-bit packed read: logical index i -> base (start + i*strand), complemented when comp */
struct wtz_seq_packed { const uint64_t *bits; int64_t start; int32_t strand; uint32_t comp;
	WTZ_HDM uint32_t at(int32_t i) const { uint32_t b = wtz_base_at(bits, (uint64_t)(start + (int64_t)i * strand)); return comp ? (3u - b) : b; } };

/* 32 logical bases [b0, b0+32) of a packed view as one word, base b0+k at bits 2k (bases at or beyond `len` read as 0):
 * two 64-bit loads and a funnel shift instead of 32 single-base extractions.  Forward views need the 2-bit groups of the
 * storage order (base i at bits ((~i)&31)*2, dna.h:78) reversed; reverse-complement views are already in ascending bit order. */
WTZ_HD uint64_t wtz_rev2bit(uint64_t x){
	x = ((x & 0x3333333333333333ULL) << 2) | ((x >> 2) & 0x3333333333333333ULL);
	x = ((x & 0x0F0F0F0F0F0F0F0FULL) << 4) | ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL);
	x = ((x & 0x00FF00FF00FF00FFULL) << 8) | ((x >> 8) & 0x00FF00FF00FF00FFULL);
	x = ((x & 0x0000FFFF0000FFFFULL) << 16) | ((x >> 16) & 0x0000FFFF0000FFFFULL);
	return (x << 32) | (x >> 32);
}
WTZ_HD uint64_t wtz_pack32(const wtz_seq_packed &s, int32_t b0, int32_t len){
	const int32_t nb = len - b0 < 32 ? len - b0 : 32;
	if(nb <= 0) return 0;
	uint64_t v;
	if(s.strand > 0){
		const uint64_t p = (uint64_t)(s.start + b0); const uint32_t o = (uint32_t)(p & 31u);
		const uint64_t w0 = s.bits[p >> 5];
		const uint64_t w1 = (o && (int32_t)(32u - o) < nb) ? s.bits[(p >> 5) + 1] : 0;
		v = o ? ((w0 << (2 * o)) | (w1 >> (64 - 2 * o))) : w0;          /* base p in the top two bits */
		v = wtz_rev2bit(v);
	} else {
		const uint64_t p = (uint64_t)(s.start - b0); const uint32_t o = (uint32_t)(p & 31u);
		const uint64_t w1 = s.bits[p >> 5];
		const uint64_t w0 = (o != 31u && (int32_t)(o + 1u) < nb) ? s.bits[(p >> 5) - 1] : 0;
		const uint32_t sh = (31u - o) * 2u;
		v = sh ? ((w1 >> sh) | (w0 << (64 - sh))) : w1;                    /* base p in the low two bits */
	}
	if(s.comp) v = ~v;
	if(nb < 32) v &= (1ULL << (2 * nb)) - 1ULL;
	return v;
}

/*
 * Scalar forms of the three banded DPs (one lane per problem).  They are the host emulation's DP, the on-device cross-check of the wave
 * kernels (WTZ_SW_CHECK) and the place where problems outside every wave envelope end up (bands of thousands of columns, targets beyond
 * the LDS words).  Written the way the wave kernels compute a row, not the way the reference's fused loop does:
 *     pass 1   diag[] : the diagonal term  H(i-1, j-1) + score(q_i, t_j)  of every band column (independent per column)
 *     pass 2   one left-to-right sweep that carries the horizontal gap state: H = max{diag, V, G},  V' = max{V+e, diag+o},  G' = max{G+e, diag+o'}
 * `feed[j]` hands H(i-1, j-1) to row i (so feed[0] is the boundary column -1), `vgap[j]` is the vertical gap state of column j; cells a
 * row does not touch keep the sentinel the band rules give them.  Trace: one byte per cell, bits 1:0 = source of H (0 diagonal, 1 vertical
 * gap, 2 horizontal gap), bit 2 = vertical gap extended, bit 3 = horizontal gap extended (the nibble of the register kernels).
 */
#define WTZ_TR_VEXT 4u
#define WTZ_TR_GEXT 8u

/* walk a byte trace back from (row r, column c): ROWBEG(i) = first band column of row i.  op: 0 = aligned pair, 1 = row only, 2 = column only */
template<typename ROWBEG, typename EMIT>
WTZ_HD void wtz_trace_walk(const uint8_t *z, int32_t n_col, int32_t &r, int32_t &c, ROWBEG rowbeg, EMIT emit){
	uint32_t state = 0;
	while(r >= 0 && c >= 0){
		const uint32_t t = z[(size_t)r * n_col + (c - rowbeg(r))];
		if(state == 0) state = t & 3u; else if(state == 1) state = (t & WTZ_TR_VEXT) ? 1u : 0u; else state = (t & WTZ_TR_GEXT) ? 2u : 0u;
		emit(state, r, c);
		if(state == 0){ r--; c--; } else if(state == 1) r--; else c--;
	}
}

/* K-sw1 (FOLLOW = false: the band is fixed around the diagonal, the row's LAST arg-max counts, kswx.h:234-335) and
 * K-sw3 (FOLLOW = true: the band centre follows the row's FIRST arg-max by +-1, kswx.h:101-232) */
template<bool FOLLOW, typename SQ, typename ST>
WTZ_HD wtz_aln_t wtz_extend_scalar(int32_t qlen, const SQ &query, int32_t tlen, const ST &target, int32_t init_score,
		int32_t W, int32_t M, int32_t X, int32_t I, int32_t D, int32_t E, int32_t T, wtz_swmem_t &mem, wtz_cigar_t &cigars, unsigned long long *cells){
	wtz_aln_t x; memset(&x, 0, sizeof x);
	cigars.n = 0;
	if(init_score < 0) init_score = 0;
	if(qlen <= 0 || tlen <= 0){ x.score = init_score; return x; }
	int32_t rows, cols, n_col;
	wtz_ext_geometry(qlen, tlen, init_score, W, M, I, D, E, T, rows, cols, n_col);
	if(!wtz_swmem_need(mem, (uint32_t)cols + 3, FOLLOW ? (uint32_t)rows + 2 : 0, (uint64_t)rows * n_col)) return x;
	int32_t *feed = mem.rh, *vgap = mem.re, *diag = mem.rm, *first_col = mem.zb;
	const int32_t open_v = I + E, open_g = D + E, NEG = -10000;
	feed[0] = init_score;                                            /* H(-1,-1) */
	for(int32_t j = 1; j <= cols; j++) feed[j] = init_score + D + E * j;       /* H(-1, j-1): a leading horizontal gap */
	for(int32_t j = 0; j <= cols; j++) vgap[j] = NEG;
	int32_t best = init_score, best_r = -1, best_c = -1;             /* running maximum over the rows */
	int32_t edge = 0, edge_r = -1, edge_c = -1;                      /* best cell on the last column / last row (the T end rule) */
	int32_t centre = 0;
	unsigned long long ncell = 0;
	for(int32_t i = 0; i < rows; i++){
		int32_t jb, je;
		if(FOLLOW){ jb = centre - W > 0 ? centre - W : 0; je = centre + W + 1 < cols ? centre + W + 1 : cols; first_col[i] = jb; }
		else      { jb = i - W > 0 ? i - W : 0; je = i + W + 1 < cols ? i + W + 1 : cols; }
		uint8_t *zi = mem.z + (size_t)i * n_col;
		ncell += (unsigned long long)(je - jb);
		/* pass 1 */
		const uint32_t qb = query.at(i);
		for(int32_t j = jb; j < je; j++) diag[j - jb] = feed[j] + ((qb == target.at(j)) ? M : X);
		/* pass 2 */
		int32_t left = (jb == 0) ? init_score + I + E * (i + 1) : NEG;       /* H(i, jb-1): column -1 is a leading vertical gap */
		int32_t g = NEG, row_max = 0, row_arg = -1;
		for(int32_t j = jb; j < je; j++){
			const int32_t m = diag[j - jb], v = vgap[j];
			uint32_t t = m >= v ? 0u : 1u;
			int32_t h = m >= v ? m : v;
			if(h < g){ t = 2u; h = g; }
			feed[j] = left; left = h;                                    /* column j+1 of the next row needs H(i, j) */
			if(FOLLOW){ if(h > row_max){ row_max = h; row_arg = j; } }    /* first arg-max, kswx.h:172 */
			else if(h >= row_max){ row_max = h; row_arg = j; }            /* last arg-max at or above 0, kswx.h:288-289 */
			const int32_t vo = m + open_v, ve = v + E;
			if(ve > vo) t |= WTZ_TR_VEXT;
			vgap[j] = ve > vo ? ve : vo;
			const int32_t go = m + open_g, ge = g + E;
			if(ge > go) t |= WTZ_TR_GEXT;
			g = ge > go ? ge : go;
			zi[j - jb] = (uint8_t)t;
		}
		feed[je] = left; vgap[je] = NEG;
		if(je == tlen && edge < left){ edge = left; edge_r = i; edge_c = je - 1; }
		if(i + 1 == qlen && edge < row_max){ edge = row_max; edge_r = i; edge_c = row_arg; }
		if(row_max > best){ best = row_max; best_r = i; best_c = row_arg; }
		else if(row_max <= 0) break;                                     /* kswx.h:185 / 302 */
		if(FOLLOW){
			/* the band moves by 0, 1 or 2 columns; what enters it from outside the previous band must read as the sentinel */
			centre++;
			if(centre < row_arg){ centre++; if(je < cols){ feed[je + 1] = NEG; vgap[je + 1] = NEG; } }
			else if(centre > row_arg) centre--;
		}
	}
	if(cells) *cells += ncell;
	if(edge > 0 && edge >= best + T){ x.score = edge; x.qe = edge_r; x.te = edge_c; }
	else { x.score = best; x.qe = best_r; x.te = best_c; }
	int32_t r = x.qe, c = x.te;
	auto emit = [&](uint32_t op, int32_t rr, int32_t cc){
		if(op == 0){ if(query.at(rr) == target.at(cc)) x.mat++; else x.mis++; } else if(op == 1) x.ins++; else x.del++;
		wtz_cigar_push(cigars, op, 1);
	};
	if(FOLLOW) wtz_trace_walk(mem.z, n_col, r, c, [&](int32_t rr){ return first_col[rr]; }, emit);
	else       wtz_trace_walk(mem.z, n_col, r, c, [&](int32_t rr){ return rr > W ? rr - W : 0; }, emit);
	if(r >= 0){ x.ins += r + 1; wtz_cigar_push(cigars, 1, (uint32_t)(r + 1)); }
	if(c >= 0){ x.del += c + 1; wtz_cigar_push(cigars, 2, (uint32_t)(c + 1)); }
	wtz_cigar_reverse(cigars.a, cigars.n);
	x.aln = x.mat + x.mis + x.ins + x.del; x.qe++; x.te++;
	return x;
}
template<typename SQ, typename ST>
WTZ_HD wtz_aln_t wtz_extend_fixed(int32_t qlen, const SQ &query, int32_t tlen, const ST &target, int32_t init_score,
		int32_t W, int32_t M, int32_t X, int32_t I, int32_t D, int32_t E, int32_t T, wtz_swmem_t &mem, wtz_cigar_t &cigars){
	return wtz_extend_scalar<false>(qlen, query, tlen, target, init_score, W, M, X, I, D, E, T, mem, cigars, (unsigned long long*)NULL);
}
template<typename SQ, typename ST>
WTZ_HD wtz_aln_t wtz_extend_shift(int32_t qlen, const SQ &query, int32_t tlen, const ST &target, int32_t init_score,
		int32_t W, int32_t M, int32_t X, int32_t I, int32_t D, int32_t E, int32_t T, wtz_swmem_t &mem, wtz_cigar_t &cigars, unsigned long long *cells){
	return wtz_extend_scalar<true>(qlen, query, tlen, target, init_score, W, M, X, I, D, E, T, mem, cigars, cells);
}

#define WTZ_MINUS_INF (-0x40000000)

/* K-sw2, ksw_global2 (ksw.c:503-586): rows run over the TARGET, band columns over the query, no early exit, the score is the corner
 * cell and the walk starts there; match M on equal bases else X (the reference passes a 4x4 matrix with exactly that content,
 * hzm_aln.h:1354); penalties positive.  cigar ops M0 / I1 (query only) / D2 (target only). */
template<typename SQ, typename ST>
WTZ_HD int32_t wtz_global_banded(int32_t qlen, const SQ &query, int32_t tlen, const ST &target, int32_t M, int32_t X,
		int32_t o_del, int32_t e_del, int32_t o_ins, int32_t e_ins, int32_t w, wtz_swmem_t &mem, wtz_cigar_t &cig){
	cig.n = 0;
	const int32_t n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	if(!wtz_swmem_need(mem, (uint32_t)qlen + 3, 0, (uint64_t)(n_col > 0 ? n_col : 0) * (uint64_t)(tlen > 0 ? tlen : 0) + 8)) return 0;
	int32_t *feed = mem.rh, *vgap = mem.re, *diag = mem.rm; uint8_t *z = mem.z;
	const int32_t open_v = o_del + e_del, open_g = o_ins + e_ins;
	/* row -1: a leading query-only gap inside the band, nothing beyond it */
	feed[0] = 0;
	for(int32_t j = 1; j <= qlen; j++) feed[j] = j <= w ? -(o_ins + e_ins * j) : WTZ_MINUS_INF;
	for(int32_t j = 0; j <= qlen; j++) vgap[j] = WTZ_MINUS_INF;
	for(int32_t i = 0; i < tlen; i++){
		const int32_t jb = i > w ? i - w : 0, je = i + w + 1 < qlen ? i + w + 1 : qlen;
		uint8_t *zi = z + (size_t)i * n_col;
		const uint32_t tb = target.at(i);
		for(int32_t j = jb; j < je; j++) diag[j - jb] = feed[j] + ((tb == query.at(j)) ? M : X);
		int32_t left = jb == 0 ? -(o_del + e_del * (i + 1)) : WTZ_MINUS_INF;        /* column -1: a leading target-only gap */
		int32_t g = WTZ_MINUS_INF;
		for(int32_t j = jb; j < je; j++){
			const int32_t m = diag[j - jb], v = vgap[j];
			uint32_t t = m >= v ? 0u : 1u;
			int32_t h = m >= v ? m : v;
			if(h < g){ t = 2u; h = g; }
			feed[j] = left; left = h;
			const int32_t vo = m - open_v, ve = v - e_del;
			if(ve > vo) t |= WTZ_TR_VEXT;
			vgap[j] = ve > vo ? ve : vo;
			const int32_t go = m - open_g, ge = g - e_ins;
			if(ge > go) t |= WTZ_TR_GEXT;
			g = ge > go ? ge : go;
			zi[j - jb] = (uint8_t)t;
		}
		feed[je] = left; vgap[je] = WTZ_MINUS_INF;
	}
	const int32_t score = feed[qlen];
	int32_t r = tlen - 1, c = (r + w + 1 < qlen ? r + w + 1 : qlen) - 1;
	wtz_trace_walk(z, n_col, r, c, [&](int32_t rr){ return rr > w ? rr - w : 0; },
		[&](uint32_t op, int32_t, int32_t){ wtz_cigar_push(cig, op == 0 ? 0u : (op == 1 ? 2u : 1u), 1); });      /* a row-only step consumes target: D */
	if(r >= 0) wtz_cigar_push(cig, 2, (uint32_t)(r + 1));
	if(c >= 0) wtz_cigar_push(cig, 1, (uint32_t)(c + 1));
	wtz_cigar_reverse(cig.a, cig.n);
	return score;
}

/* hzm_aln.h:278-314: run-by-run alignment of a matched z-mer (homopolymer length differences -> I/D) */
template<typename S1, typename S2>
WTZ_HD wtz_aln_t wtz_align_zmer(const S1 &pb1, uint32_t len1, const S2 &pb2, uint32_t len2, int32_t M, int32_t I, int32_t D, int32_t E, wtz_cigar_t &cigars){
	wtz_aln_t x, zero; memset(&zero, 0, sizeof zero); x = zero;
	uint32_t s0 = 0, s1 = 0, e0, e1, l0, l1;
	while(s0 < len1 || s1 < len2){
		if(pb1.at((int32_t)s0) != pb2.at((int32_t)s1)) return zero;
		e0 = s0 + 1; while(e0 < len1 && pb1.at((int32_t)e0) == pb1.at((int32_t)s0)) e0++;
		e1 = s1 + 1; while(e1 < len2 && pb2.at((int32_t)e1) == pb2.at((int32_t)s1)) e1++;
		l0 = e0 - s0; l1 = e1 - s1;
		if(l0 < l1){
			x.aln += l1; x.mat += l0; x.ins += l1 - l0; x.score += (int32_t)l0 * M + I + (int32_t)(l1 - l0) * E;
			wtz_cigar_push(cigars, 0, l0); wtz_cigar_push(cigars, 1, l1 - l0);
		} else if(l0 == l1){
			x.aln += l0; x.mat += l0; x.score += (int32_t)l0 * M;
			wtz_cigar_push(cigars, 0, l0);
		} else {
			x.aln += l0; x.mat += l1; x.del += l0 - l1; x.score += (int32_t)l1 * M + D + (int32_t)(l0 - l1) * E;
			wtz_cigar_push(cigars, 0, l1); wtz_cigar_push(cigars, 2, l0 - l1);
		}
		s0 = e0; s1 = e1;
	}
	x.te = x.mat + x.del; x.qe = x.mat + x.ins;
	return x;
}

/* Reads as the aligner sees them: pb1 = query read forward; pb2 = candidate, reverse-complemented when dir (wtzmo.c:1011-1013) */
struct wtz_readview { const uint64_t *bits; uint64_t off; uint32_t len; uint32_t rev;
	WTZ_HDM wtz_seq_packed sub(int32_t from, int32_t strand) const {      /* logical position `from`, walking by strand */
		wtz_seq_packed s; s.bits = bits;
		if(!rev){ s.start = (int64_t)off + from; s.strand = strand; s.comp = 0; }
		else    { s.start = (int64_t)off + (int64_t)len - 1 - from; s.strand = -strand; s.comp = 1; }
		return s;
	} };

/* A9 (hzm_aln.h:1247-1302): x accumulates along the anchors of one window; cigar appended to `cigar` */
WTZ_HD wtz_aln_t wtz_align_window(const wtz_readview &pb1, const wtz_readview &pb2, const wtz_win_t &win, const wtz_zhit_t *anchors,
		wtz_cigar_t &cigar, wtz_swmem_t &mem, wtz_cigar_t &tmp, const wtz_params_t *P){
	wtz_aln_t x, y; memset(&x, 0, sizeof x);
	const int32_t M = P->M, X = P->X, I = P->O, D = P->O, E = P->E, T = P->T;
	for(uint32_t i = win.anchors[0]; i < win.anchors[1]; i++){
		const wtz_zhit_t p = anchors[i];
		const int32_t off1 = (int32_t)ZH_OFF1(p), off2 = (int32_t)ZH_OFF2(p);
		if(x.aln == 0){ x.tb = x.te = off1; x.qb = x.qe = off2; }
		if(off1 < x.te) continue;
		if(off2 < x.qe) continue;
		tmp.n = 0;
		y = wtz_extend_fixed(off2 - x.qe, pb2.sub(x.qe, 1), off1 - x.te, pb1.sub(x.te, 1), x.score, P->w, M, X, I, D, E, T, mem, tmp);
		x.score = y.score;
		x.aln += y.aln; x.mat += y.mat; x.mis += y.mis; x.ins += y.ins; x.del += y.del;
		x.te += y.te; x.qe += y.qe;
		if(x.te < off1){ x.del += off1 - x.te; x.aln += off1 - x.te; wtz_cigar_push(tmp, 2, (uint32_t)(off1 - x.te)); x.te = off1; }
		if(x.qe < off2){ x.ins += off2 - x.qe; x.aln += off2 - x.qe; wtz_cigar_push(tmp, 1, (uint32_t)(off2 - x.qe)); x.qe = off2; }
		wtz_cigar_concat(cigar, tmp.a, tmp.n);
		tmp.n = 0;
		y = wtz_align_zmer(pb1.sub(off1, 1), ZH_LEN1(p), pb2.sub(off2, 1), ZH_LEN2(p), M, I, D, E, tmp);
		if(y.aln == 0) return x;
		x.score += y.score;
		x.aln += y.aln; x.mat += y.mat; x.mis += y.mis; x.ins += y.ins; x.del += y.del;
		x.te += y.te; x.qe += y.qe;
		wtz_cigar_concat(cigar, tmp.a, tmp.n);
	}
	return x;
}

#endif
