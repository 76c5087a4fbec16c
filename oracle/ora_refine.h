/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see ora_util.h).
 *
 * ora_refine.h — the optional `-n` pass (A11): re-alignment of a finished overlap inside a band drawn around its own
 * CIGAR.  Restates reference kswx.h:483-659 (kswx_refine_alignment), called from wtzmo.c:1031-1034 with
 * query = candidate read (oriented), target = query read, W = -w, I = D = -O.
 *
 * What the reference does, in order:
 *   1. the CIGAR gives ql / tl (483-508); an empty side returns the null alignment and an empty CIGAR;
 *   2. per query row a half-width zw[]: W on M rows, W + len on the rows of an insertion of length len (536-549; the
 *      `zw[qx] += len` of a deletion lands on a row that the next M/I operation overwrites, so it has no effect);
 *   3. widening around indels: an insertion or deletion of length len adds len - j to the rows j < len before it and
 *      after it (551-571: for an insertion "after" starts at its last row);
 *   4. band [zb, ze) per row = diagonal position of the row +- zw, clamped to [0, tl] (573-599), then zb made
 *      non-decreasing and ze non-increasing from the end (601-611);
 *   5. global DP over the rows with the extension recurrence (-10000 sentinels; gaps open from the diagonal value), the
 *      row arrays being reused across rows WITHOUT re-initialisation: rh[j] holds H(i-1, j-1) only where the previous
 *      row wrote it, older values otherwise (613-641) - kept verbatim by using the same arrays;
 *   6. score = rh[tl]; traceback from (ql-1, tl-1) (643-657).
 * The reference stores the trace in a ql x tl byte matrix indexed z[i*tl + (j - zb[i])]; only band cells are ever
 * written or read, so this restatement keeps one row offset per row instead.
 */
#ifndef ORA_REFINE_H
#define ORA_REFINE_H

#include "ora_sw.h"

typedef struct { vec_i32 rh, re, zb, ze, zw; vec_u8 z; vec_u64 zoff; } ora_refmem_t;

static inline void ora_refmem_free(ora_refmem_t *m){
	vec_i32_free(&m->rh); vec_i32_free(&m->re); vec_i32_free(&m->zb); vec_i32_free(&m->ze); vec_i32_free(&m->zw); vec_u8_free(&m->z); vec_u64_free(&m->zoff);
}

/* query / target: one base per byte; cigars: the stitched CIGAR (M0/I1/D2); out: the refined CIGAR */
static ora_aln_t ora_refine_alignment(const uint8_t *query, int qb, const uint8_t *target, int tb, int W, int M, int X, int I, int D, int E,
		const uint32_t *cigars, size_t n_cigar, ora_refmem_t *mem, vec_u32 *out){
	ora_aln_t y; memset(&y, 0, sizeof y);
	int qe = qb, te = tb, ql, tl, qx, tx, op, len, zb_, ze_;
	long long i, j;
	out->n = 0;
	for(i = 0; i < (long long)n_cigar; i++){
		op = (int)(cigars[i] & 0xFu); len = (int)(cigars[i] >> 4);
		if(op == 0){ qe += len; te += len; } else if(op == 1) qe += len; else te += len;
	}
	ql = qe - qb; tl = te - tb;
	if(ql == 0 || tl == 0) return y;                     /* KSWX_NULL */
	vec_i32_reserve(&mem->rh, (size_t)tl + 2); vec_i32_reserve(&mem->re, (size_t)tl + 2);
	vec_i32_reserve(&mem->zb, (size_t)ql + 2); vec_i32_reserve(&mem->ze, (size_t)ql + 2); vec_i32_reserve(&mem->zw, (size_t)ql + 2);
	vec_u64_reserve(&mem->zoff, (size_t)ql + 2);
	int *rh = mem->rh.a, *re = mem->re.a, *zb = mem->zb.a, *ze = mem->ze.a, *zw = mem->zw.a;
	for(i = 0; i < ql + 2; i++) zw[i] = 0;
	/* basic half-width */
	qx = 0;
	for(i = 0; i < (long long)n_cigar; i++){
		op = (int)(cigars[i] & 0xFu); len = (int)(cigars[i] >> 4);
		if(op == 0){ for(j = 0; j < len; j++) zw[qx++] = W; }
		else if(op == 1){ for(j = 0; j < len; j++) zw[qx++] = W + len; }
		/* op 2: see note 2 in the header */
	}
	/* widening around indels */
	qx = 0;
	for(i = 0; i < (long long)n_cigar; i++){
		op = (int)(cigars[i] & 0xFu); len = (int)(cigars[i] >> 4);
		if(op == 0) qx += len;
		else if(op == 1){
			for(j = 1; j < len && j < qx; j++) zw[qx - j] += len - (int)j;
			qx += len - 1;
			for(j = 1; j < len && j + qx < ql; j++) zw[qx + j] += len - (int)j;
			qx++;
		} else {
			for(j = 1; j < len && j < qx; j++) zw[qx - j] += len - (int)j;
			for(j = 1; j < len && j + qx < ql; j++) zw[qx + j] += len - (int)j;
		}
	}
	/* band per row */
	tx = qx = 0;
	for(i = 0; i < (long long)n_cigar; i++){
		op = (int)(cigars[i] & 0xFu); len = (int)(cigars[i] >> 4);
		if(op == 0 || op == 1){
			for(j = 0; j < len; j++){
				zb_ = tx - zw[qx]; if(zb_ < 0) zb_ = 0;
				ze_ = tx + 1 + zw[qx]; if(ze_ > tl) ze_ = tl;
				zb[qx] = zb_; ze[qx] = ze_;
				if(op == 0) tx++;
				qx++;
			}
		} else tx += len;
	}
	zb_ = 0;
	for(i = 0; i < ql; i++){ if(zb[i] < zb_) zb[i] = zb_; else if(zb[i] > zb_) zb_ = zb[i]; }
	ze_ = tl;
	for(i = ql - 1; i >= 0; i--){ if(ze[i] > ze_) ze[i] = ze_; else if(ze[i] < ze_) ze_ = ze[i]; }
	/* trace rows: band cells only */
	{
		uint64_t tot = 0;
		for(i = 0; i < ql; i++){ mem->zoff.a[i] = tot; tot += (uint64_t)(ze[i] > zb[i] ? ze[i] - zb[i] : 0); }
		vec_u8_reserve(&mem->z, (size_t)tot + 8);
	}
	uint8_t *z = mem->z.a;
	rh[0] = 0;
	for(i = 1; i <= tl; i++) rh[i] = -10000;
	for(i = 0; i <= tl; i++) re[i] = -10000;
	for(i = 0; i < ql; i++){
		const int qc = query[i + qb];
		int h1 = -10000, f = -10000, h, m, e, t; uint8_t d;
		uint8_t *zi = z + mem->zoff.a[i];
		for(j = zb[i]; j < ze[i]; j++){
			m = rh[j] + ((qc == target[j + tb]) ? M : X);
			rh[j] = h1;
			e = re[j];
			if(m >= e){ d = 0; h = m; } else { d = 1; h = e; }
			if(h < f){ d = 2; h = f; }
			h1 = h;
			t = m + I + E; e = e + E; if(e > t) d |= 1 << 2; else e = t;
			re[j] = e;
			t = m + D + E; f = f + E; if(f > t) d |= 2 << 4; else f = t;
			zi[j - zb[i]] = d;
		}
		rh[j] = h1; re[j] = -10000;
	}
	y.qb = qb; y.qe = qe; y.tb = tb; y.te = te;
	y.score = rh[tl];
	{
		uint32_t d = 0;
		i = ql - 1; j = tl - 1;
		while(i >= 0 && j >= 0){
			d = (z[mem->zoff.a[i] + (uint64_t)(j - zb[i])] >> (d << 1)) & 0x03;
			if(d == 0){ if(query[i + y.qb] == target[j + y.tb]) y.mat++; else y.mis++; i--; j--; }
			else if(d == 1){ i--; y.ins++; }
			else { j--; y.del++; }
			ora_cigar_push(out, d, 1);
		}
		if(i >= 0){ y.ins += (int)i + 1; ora_cigar_push(out, 1, (uint32_t)(i + 1)); }
		if(j >= 0){ y.del += (int)j + 1; ora_cigar_push(out, 2, (uint32_t)(j + 1)); }
		ora_cigar_reverse(out);
	}
	y.aln = y.mat + y.mis + y.ins + y.del;
	return y;
}

#endif
