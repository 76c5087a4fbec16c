/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see ora_util.h).
 *
 * ora_sw.h — the three banded dynamic programs of the zmo engine and CIGAR plumbing.
 * Restates:
 *   - CIGAR push / concat / text      reference kswx.h:39-52, 1093-1120
 *   - fixed-band extension  (K-sw1)   reference kswx.h:234-335  kswx_extend_align_core
 *   - shifting-band extension (K-sw3) reference kswx.h:101-232  kswx_extend_align_shift_core
 *   - global banded alignment (K-sw2) reference ksw.c:503-586   ksw_global2
 *
 * Recurrence shared by the two extension kernels (all int32; -10000 is the reference's
 * "minus infinity" and is part of the semantics):
 *   m(i,j)   = H(i-1,j-1) + S(q_i,t_j)
 *   H(i,j)   = max{m, E(i,j), F(i,j)}            ties: m over E, {m,E} over F
 *   E(i+1,j) = max{E(i,j)+e, m(i,j)+I+e}         bit 2 of the trace byte set when E extends
 *   F(i,j+1) = max{F(i,j)+e, m(i,j)+D+e}         bit 5 of the trace byte set when F extends
 * Note gaps open from m (the diagonal value), not from H.
 */
#ifndef ORA_SW_H
#define ORA_SW_H

#include "ora_util.h"

typedef struct { int score, tb, te, qb, qe, aln, mat, mis, ins, del; } ora_aln_t;   /* kswx_t */

static inline void ora_cigar_push(vec_u32 *c, uint32_t op, uint32_t len){            /* kswx.h:39-44 */
	if(len == 0) return;
	if(c->n && (c->a[c->n - 1] & 0xF) == op) c->a[c->n - 1] += len << 4;
	else vec_u32_push(c, (len << 4) | op);
}

static inline void ora_cigar_concat(vec_u32 *c, const uint32_t *src, size_t n){      /* kswx.h:46-52 */
	if(n == 0) return;
	if(c->n && (c->a[c->n - 1] & 0x0F) == (src[0] & 0x0Fu)){
		c->a[c->n - 1] += src[0] & 0xFFFFFFF0u;
		vec_u32_append(c, src + 1, n - 1);
	} else vec_u32_append(c, src, n);
}

static inline void ora_cigar_reverse(vec_u32 *c){
	for(size_t i = 0; i < c->n / 2; i++){ uint32_t t = c->a[i]; c->a[i] = c->a[c->n - 1 - i]; c->a[c->n - 1 - i] = t; }
}

static inline void ora_cigar_text(vec_u8 *out, const uint32_t *c, size_t n){          /* kswx.h:1093-1120 */
	char buf[24];
	for(size_t i = 0; i < n; i++){
		uint32_t op = c[i] & 0xF, len = c[i] >> 4;
		if(len == 0) continue;
		if(op > 2){ fprintf(stderr, "oracle: bad cigar op %u\n", op); exit(1); }
		int k = snprintf(buf, sizeof buf, "%u%c", len, "MID"[op]);
		vec_u8_append(out, (uint8_t*)buf, (size_t)k);
	}
}

typedef struct { vec_i32 rh, re, zb; vec_u8 z; } ora_swmem_t;

/* common prologue of both extension kernels: clamp the band, derive ql/tl/n_col */
static inline void ora_ext_geometry(int qlen, int tlen, int init_score, int *W, int M, int I, int D, int E, int T, int *ql, int *tl, int *n_col){
	int w = *W;
	if(w > 0){
		int max = ((qlen < tlen) ? qlen : tlen) * M + init_score + (-T);
		int max_gap = (max + ((I > D) ? I : D)) / (-E) + 1;
		if(max_gap < 1) max_gap = 1;
		if(w > max_gap) w = max_gap;
	} else w = -w;
	w = ORA_MIN(w, ORA_MAX(qlen, tlen));
	if(qlen < tlen){ if(qlen + w < tlen){ *ql = qlen; *tl = qlen + w; } else { *ql = qlen; *tl = tlen; } }
	else           { if(tlen + w < qlen){ *tl = tlen; *ql = tlen + w; } else { *tl = tlen; *ql = qlen; } }
	*n_col = (*tl < 2 * w + 1) ? *tl : 2 * w + 1;
	*W = w;
}

/* shared traceback: zrow(i) gives the first column stored for row i */
#define ORA_EXT_TRACEBACK(ZROW) do { \
	int i_ = x.qe, j_ = x.te; uint8_t d_ = 0; \
	while(i_ >= 0 && j_ >= 0){ \
		d_ = (mem->z.a[(size_t)i_ * n_col + (j_ - (ZROW))] >> (d_ << 1)) & 0x03; \
		if(d_ == 0){ if(query[i_ * strand] == target[j_ * strand]) x.mat++; else x.mis++; i_--; j_--; } \
		else if(d_ == 1){ i_--; x.ins++; } \
		else { j_--; x.del++; } \
		ora_cigar_push(cigars, d_, 1); \
	} \
	if(i_ >= 0){ x.ins += i_ + 1; ora_cigar_push(cigars, 1, (uint32_t)(i_ + 1)); } \
	if(j_ >= 0){ x.del += j_ + 1; ora_cigar_push(cigars, 2, (uint32_t)(j_ + 1)); } \
	ora_cigar_reverse(cigars); \
	x.aln = x.mat + x.mis + x.ins + x.del; x.qe++; x.te++; \
} while(0)

/* K-sw1, kswx.h:234-335. query/target are indexed as ptr[i*strand]. cigars is cleared. */
static ora_aln_t ora_extend_fixed(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int strand, int init_score,
		int W, int M, int X, int I, int D, int E, int T, ora_swmem_t *mem, vec_u32 *cigars){
	ora_aln_t x; memset(&x, 0, sizeof x);
	int ql, tl, n_col, i, j, jb, je, h1, h, m, e, f, t;
	int max, mi, mj, imax, mj2, gmax, gi, gj;
	uint8_t d;
	if(init_score < 0) init_score = 0;
	if(qlen <= 0 || tlen <= 0){ x.score = init_score; return x; }
	ora_ext_geometry(qlen, tlen, init_score, &W, M, I, D, E, T, &ql, &tl, &n_col);
	mem->rh.n = mem->re.n = 0; vec_i32_reserve(&mem->rh, (size_t)tl + 2); vec_i32_reserve(&mem->re, (size_t)tl + 2);
	mem->z.n = 0; vec_u8_reserve(&mem->z, (size_t)ql * n_col + 8);
	int *rh = mem->rh.a, *re = mem->re.a;
	rh[0] = init_score; rh[1] = init_score + D + E;
	for(j = 2; j <= tl; j++) rh[j] = rh[j - 1] + E;
	for(j = 0; j <= tl; j++) re[j] = -10000;
	max = init_score; mi = -1; mj = -1; gmax = 0; gi = -1; gj = -1;
#ifdef ORA_STATS
	int st_minrow = 1 << 30, st_break = 0, st_gcand = -(1 << 30);
#endif
	for(i = 0; i < ql; i++){
		jb = i - W; if(jb < 0) jb = 0;
		je = i + W + 1; if(je > tl) je = tl;
		h1 = (jb == 0) ? init_score + I + E * (i + 1) : -10000;
		uint8_t *zi = mem->z.a + (size_t)i * n_col;
		imax = 0; mj2 = -1; f = -10000;
#ifdef ORA_STATS
		int st_rowmax = -(1 << 30);
#endif
		for(j = jb; j < je; j++){
			m = rh[j] + ((query[i * strand] == target[j * strand]) ? M : X);
			rh[j] = h1;
			e = re[j];
			d = m >= e ? 0 : 1;
			h = m >= e ? m : e;
			d = h >= f ? d : 2;
			h = h >= f ? h : f;
			h1 = h;
#ifdef ORA_STATS
			if(h > st_rowmax) st_rowmax = h;
#endif
			mj2  = imax > h ? mj2 : j;      /* last arg-max: ties take the larger j (kswx.h:288-289); K-sw3 keeps the first */
			imax = imax > h ? imax : h;
			t = m + I + E; e = e + E;
			d |= e > t ? 1 << 2 : 0; e = e > t ? e : t;
			re[j] = e;
			t = m + D + E; f = f + E;
			d |= f > t ? 2 << 4 : 0; f = f > t ? f : t;
			zi[j - jb] = d;
		}
		rh[j] = h1; re[j] = -10000;
#ifdef ORA_STATS
		if(st_rowmax - init_score < st_minrow) st_minrow = st_rowmax - init_score;
		if(j == tlen && h1 - init_score > st_gcand) st_gcand = h1 - init_score;
		if(i + 1 == qlen && st_rowmax - init_score > st_gcand) st_gcand = st_rowmax - init_score;
#endif
		if(j == tlen && gmax < h1){ gmax = h1; gi = i; gj = j - 1; }
		if(i + 1 == qlen && gmax < imax){ gmax = imax; gi = i; gj = mj2; }
		if(imax > max){ max = imax; mi = i; mj = mj2; }
		else if(imax <= 0){
#ifdef ORA_STATS
			st_break = 1;
#endif
			break; }
	}
#ifdef ORA_STATS
	/* shape, init, min over rows of (row max - init), early break, best end-candidate - init (or -inf), max - init */
	fprintf(stderr, "SW1DP\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n", ql, tl, n_col, init_score, st_minrow, st_break, st_gcand, max - init_score);
#endif
	if(gmax > 0 && gmax >= max + T){ x.score = gmax; x.qe = gi; x.te = gj; }
	else { x.score = max; x.qe = mi; x.te = mj; }
	cigars->n = 0;
	ORA_EXT_TRACEBACK((i_ > W ? i_ - W : 0));
	return x;
}

/* K-sw3, kswx.h:101-232: the band centre follows each row's arg-max by at most one column */
static ora_aln_t ora_extend_shift(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int strand, int init_score,
		int W, int M, int X, int I, int D, int E, int T, ora_swmem_t *mem, vec_u32 *cigars){
	ora_aln_t x; memset(&x, 0, sizeof x);
	int ql, tl, n_col, i, j, jb, je, h1, c, h, m, e, f, t;
	int max, mi, mj, imax, mj2, gmax, gi, gj;
	uint8_t d;
	cigars->n = 0;
	if(init_score < 0) init_score = 0;
	if(qlen <= 0 || tlen <= 0){ x.score = init_score; return x; }
	ora_ext_geometry(qlen, tlen, init_score, &W, M, I, D, E, T, &ql, &tl, &n_col);
	mem->rh.n = mem->re.n = mem->zb.n = 0;
	vec_i32_reserve(&mem->rh, (size_t)tl + 3); vec_i32_reserve(&mem->re, (size_t)tl + 3); vec_i32_reserve(&mem->zb, (size_t)ql + 2);
	mem->z.n = 0; vec_u8_reserve(&mem->z, (size_t)ql * n_col + 8);
	int *rh = mem->rh.a, *re = mem->re.a, *zb = mem->zb.a;
	rh[0] = init_score; rh[1] = init_score + D + E;
	for(j = 2; j <= tl; j++) rh[j] = rh[j - 1] + E;
	for(j = 0; j <= tl; j++) re[j] = -10000;
	max = init_score; mi = -1; mj = -1; gmax = 0; gi = -1; gj = -1;
	jb = 0; je = tl;
	for(i = c = 0; i < ql; i++){
		if(jb < c - W) jb = c - W;
		if(je > c + W + 1) je = c + W + 1;
		if(je > tl) je = tl;
		h1 = (jb == 0) ? init_score + I + E * (i + 1) : -10000;
		uint8_t *zi = mem->z.a + (size_t)i * n_col;
		zb[i] = jb;
		imax = 0; mj2 = -1; f = -10000;
		for(j = jb; j < je; j++){
			m = rh[j] + ((query[i * strand] == target[j * strand]) ? M : X);
			rh[j] = h1;
			e = re[j];
			if(m >= e){ d = 0; h = m; } else { d = 1; h = e; }
			if(h < f){ d = 2; h = f; }
			h1 = h;
			if(h > imax){ imax = h; mj2 = j; }
			t = m + I + E; e = e + E;
			if(e > t) d |= 1 << 2; else e = t;
			re[j] = e;
			t = m + D + E; f = f + E;
			if(f > t) d |= 2 << 4; else f = t;
			zi[j - jb] = d;
		}
		rh[j] = h1; re[j] = -10000;
		if(j == tlen && gmax < h1){ gmax = h1; gi = i; gj = j - 1; }
		if(i + 1 == qlen && gmax < imax){ gmax = imax; gi = i; gj = mj2; }
		if(imax > max){ max = imax; mi = i; mj = mj2; }
		else if(imax <= 0) break;
		c++;
		if(c < mj2){ c++; if(je < tl){ rh[je + 1] = -10000; re[je + 1] = -10000; } }
		else if(c > mj2){ c--; if(jb){ rh[jb - 1] = -10000; re[jb - 1] = -10000; } }
		jb = 0; je = tl;
	}
	if(gmax > 0 && gmax >= max + T){ x.score = gmax; x.qe = gi; x.te = gj; }
	else { x.score = max; x.qe = mi; x.te = mj; }
	ORA_EXT_TRACEBACK(zb[i_]);
	return x;
}

#define ORA_MINUS_INF (-0x40000000)

/* K-sw2, ksw.c:503-586. mat is 4x4; (o_del,e_del,o_ins,e_ins) are positive penalties.
 * Target is the outer loop. Returns the score of the last cell; cigar (ops M0/I1/D2) appended to *cig (cleared). */
static int ora_global_banded(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
		int o_del, int e_del, int o_ins, int e_ins, int w, ora_swmem_t *mem, vec_u32 *cig){
	int i, j, k, oe_del = o_del + e_del, oe_ins = o_ins + e_ins, score, n_col;
	cig->n = 0;
	n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	mem->z.n = 0; vec_u8_reserve(&mem->z, (size_t)(n_col > 0 ? n_col : 0) * (size_t)(tlen > 0 ? tlen : 0) + 8);
	mem->rh.n = mem->re.n = 0; vec_i32_reserve(&mem->rh, (size_t)qlen + 2); vec_i32_reserve(&mem->re, (size_t)qlen + 2);
	int *H = mem->rh.a, *Ev = mem->re.a;
	uint8_t *z = mem->z.a;
	H[0] = 0; Ev[0] = ORA_MINUS_INF;
	for(j = 1; j <= qlen && j <= w; ++j){ H[j] = -(o_ins + e_ins * j); Ev[j] = ORA_MINUS_INF; }
	for(; j <= qlen; ++j) H[j] = Ev[j] = ORA_MINUS_INF;
	for(i = 0; i < tlen; ++i){
		int32_t f = ORA_MINUS_INF, h1, beg, end, t;
		uint8_t *zi = &z[(size_t)i * n_col];
		beg = i > w ? i - w : 0;
		end = i + w + 1 < qlen ? i + w + 1 : qlen;
		h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : ORA_MINUS_INF;
		for(j = beg; j < end; ++j){
			int32_t h, m = H[j], e = Ev[j];
			uint8_t d;
			H[j] = h1;
			m += mat[target[i] * 4 + query[j]];
			d = m >= e ? 0 : 1;
			h = m >= e ? m : e;
			d = h >= f ? d : 2;
			h = h >= f ? h : f;
			h1 = h;
			t = m - oe_del; e -= e_del;
			d |= e > t ? 1 << 2 : 0; e = e > t ? e : t;
			Ev[j] = e;
			t = m - oe_ins; f -= e_ins;
			d |= f > t ? 2 << 4 : 0; f = f > t ? f : t;
			zi[j - beg] = d;
		}
		H[end] = h1; Ev[end] = ORA_MINUS_INF;
	}
	score = H[qlen];
	{
		int which = 0;
		i = tlen - 1; k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
		while(i >= 0 && k >= 0){
			which = z[(size_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
			if(which == 0){ ora_cigar_push(cig, 0, 1); --i; --k; }
			else if(which == 1){ ora_cigar_push(cig, 2, 1); --i; }
			else { ora_cigar_push(cig, 1, 1); --k; }
		}
		if(i >= 0) ora_cigar_push(cig, 2, (uint32_t)(i + 1));
		if(k >= 0) ora_cigar_push(cig, 1, (uint32_t)(k + 1));
		ora_cigar_reverse(cig);
	}
	return score;
}

#endif
