/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see ora_util.h).
 *
 * ora_align.h — seed-anchored window alignment (A9) and window stitching (A10).
 * Restates:
 *   - hz_align_hzmo               reference hzm_aln.h:278-314  (run-by-run alignment of a matched z-mer)
 *   - fast_seeds_align_hzmo       reference hzm_aln.h:1247-1302
 *   - global_align_regs_hzmo      reference hzm_aln.h:1345-1486
 */
#ifndef ORA_ALIGN_H
#define ORA_ALIGN_H

#include "ora_window.h"
#include "ora_sw.h"

typedef struct { ora_aln_t x; uint32_t cigar_off, cigar_len; } ora_reg_t;   /* aln_reg_t */
ORA_VEC(vec_reg, ora_reg_t)

/* hzm_aln.h:278-314 */
static ora_aln_t ora_align_zmer(const uint8_t *pb1, uint32_t len1, const uint8_t *pb2, uint32_t len2, int M, int I, int D, int E, vec_u32 *cigars){
	ora_aln_t x, zero; memset(&zero, 0, sizeof zero); x = zero;
	uint32_t s0 = 0, s1 = 0, e0, e1, l0, l1;
	while(s0 < len1 || s1 < len2){
		if(pb1[s0] != pb2[s1]) return zero;
		e0 = s0 + 1; while(e0 < len1 && pb1[e0] == pb1[s0]) e0++;
		e1 = s1 + 1; while(e1 < len2 && pb2[e1] == pb2[s1]) e1++;
		l0 = e0 - s0; l1 = e1 - s1;
		if(l0 < l1){
			x.aln += l1; x.mat += l0; x.ins += l1 - l0;
			x.score += (int)l0 * M + I + (int)(l1 - l0) * E;
			ora_cigar_push(cigars, 0, l0); ora_cigar_push(cigars, 1, l1 - l0);
		} else if(l0 == l1){
			x.aln += l0; x.mat += l0; x.score += (int)l0 * M;
			ora_cigar_push(cigars, 0, l0);
		} else {
			x.aln += l0; x.mat += l1; x.del += l0 - l1;
			x.score += (int)l1 * M + D + (int)(l0 - l1) * E;
			ora_cigar_push(cigars, 0, l1); ora_cigar_push(cigars, 2, l0 - l1);
		}
		s0 = e0; s1 = e1;
	}
	x.te = x.mat + x.del;
	x.qe = x.mat + x.ins;
	return x;
}

/* A9, hzm_aln.h:1247-1302. pb1 = query read (target axis, t*), pb2 = candidate (q*). */
static ora_aln_t ora_align_window(const uint8_t *pb1, const uint8_t *pb2, const ora_win_t *win, const ora_zhit_t *anchors,
		vec_u32 *cigar, ora_swmem_t *mem, vec_u32 *tmp_cigar, int w, int M, int X, int I, int D, int E, int T){
	ora_aln_t x, y; memset(&x, 0, sizeof x);
	for(uint32_t i = win->anchors[0]; i < win->anchors[1]; i++){
		const ora_zhit_t *p = &anchors[i];
		if(x.aln == 0){ x.tb = x.te = (int)p->off1; x.qb = x.qe = (int)p->off2; }
		if((int)p->off1 < x.te) continue;
		if((int)p->off2 < x.qe) continue;
		tmp_cigar->n = 0;
#ifdef ORA_STATS       /* analysis build only (tools/analysis): shape of every K-sw1 problem */
		fprintf(stderr, "SW1\t%u\t%d\t%d\t%d\n", win->anchors[1] - win->anchors[0], (int)p->off2 - x.qe, (int)p->off1 - x.te, x.score);
#endif
		y = ora_extend_fixed((int)p->off2 - x.qe, pb2 + x.qe, (int)p->off1 - x.te, pb1 + x.te, 1, x.score, w, M, X, I, D, E, T, mem, tmp_cigar);
		x.score = y.score;
		x.aln += y.aln; x.mat += y.mat; x.mis += y.mis; x.ins += y.ins; x.del += y.del;
		x.te += y.te; x.qe += y.qe;
		if(x.te < (int)p->off1){
			x.del += (int)p->off1 - x.te; x.aln += (int)p->off1 - x.te;
			ora_cigar_push(tmp_cigar, 2, (uint32_t)((int)p->off1 - x.te));
			x.te = (int)p->off1;
		}
		if(x.qe < (int)p->off2){
			x.ins += (int)p->off2 - x.qe; x.aln += (int)p->off2 - x.qe;
			ora_cigar_push(tmp_cigar, 1, (uint32_t)((int)p->off2 - x.qe));
			x.qe = (int)p->off2;
		}
		ora_cigar_concat(cigar, tmp_cigar->a, tmp_cigar->n);
		tmp_cigar->n = 0;
		y = ora_align_zmer(pb1 + p->off1, p->len1, pb2 + p->off2, p->len2, M, I, D, E, tmp_cigar);
		if(y.aln == 0) return x;
		x.score += y.score;
		x.aln += y.aln; x.mat += y.mat; x.mis += y.mis; x.ins += y.ins; x.del += y.del;
		x.te += y.te; x.qe += y.qe;
		ora_cigar_concat(cigar, tmp_cigar->a, tmp_cigar->n);
	}
	return x;
}

/* A10, hzm_aln.h:1345-1486. len1/pb1 = query read (t axis), len2/pb2 = candidate (q axis). */
static ora_aln_t ora_stitch_windows(int len1, int len2, const ora_reg_t *regs, size_t nreg, const int esti_regs[2],
		const uint8_t *pb1, const uint8_t *pb2, const uint32_t *cigar_cache, vec_u32 *cigar, ora_swmem_t *mem, vec_u32 *tmp_cigar,
		int W, int ew, int _w, int M, int X, int I, int D, int E, int T){
	ora_aln_t x, y; memset(&x, 0, sizeof x); memset(&y, 0, sizeof y);
	int8_t matrix[16];
	int w, max_gap, init_score, score;
	cigar->n = 0;
	if(nreg == 0) return x;
	for(int k = 0; k < 16; k++) matrix[k] = ((k % 4) == (k / 4)) ? (int8_t)M : (int8_t)X;
	init_score = 100 * M;
	const ora_reg_t *reg2 = &regs[0], *reg1;
	x = reg2->x;
	if(x.qb && x.tb){
		w = ew;
		max_gap = ((ORA_MIN(x.qb, x.tb) * M + x.score + init_score + (-T)) + (I < D ? D : I)) / (-E) + 1;
		if(max_gap < w) max_gap = w;
		for(;;){
			tmp_cigar->n = 0;
			y = ora_extend_shift(x.qb, pb2 + x.qb - 1, x.tb, pb1 + x.tb - 1, -1, x.score + init_score, -w, M, X, I, D, E, T, mem, tmp_cigar);
			if(y.qe == x.qb || y.te == x.tb) break;
			if(x.tb - y.te <= esti_regs[0]) break;
			if(w >= ew || w >= max_gap) break;
			w <<= 1;
		}
		x.score = y.score - init_score;
		x.aln += y.aln; x.mat += y.mat; x.mis += y.mis; x.ins += y.ins; x.del += y.del;
		x.qb -= y.qe; x.tb -= y.te;
		ora_cigar_reverse(tmp_cigar);
		ora_cigar_concat(cigar, tmp_cigar->a, tmp_cigar->n);
	}
	ora_cigar_concat(cigar, cigar_cache + reg2->cigar_off, reg2->cigar_len);
	reg1 = reg2;
	for(size_t i = 1; i < nreg; i++){
		reg2 = &regs[i];
		const uint8_t *q = pb2 + reg1->x.qe, *t = pb1 + reg1->x.te;
		int dq = reg2->x.qb - reg1->x.qe, dt = reg2->x.tb - reg1->x.te;
		w = _w;
		for(;;){
			if(w < ORA_ABSDIFF(dq, dt)){ w <<= 1; continue; }
			score = ora_global_banded(dq, q, dt, t, matrix, -I, -E, -D, -E, w, mem, tmp_cigar);
			if(score < 0 && w < W && w < ORA_MAX(dq, dt)) w <<= 1;
			else break;
		}
		x.score += score;
		x.qe = reg2->x.qb; x.te = reg2->x.tb;
		int x1 = 0, x2 = 0;
		for(size_t idx = 0; idx < tmp_cigar->n; idx++){
			int op = tmp_cigar->a[idx] & 0xF, len = (int)(tmp_cigar->a[idx] >> 4);
			x.aln += len;
			switch(op){
				case 0: for(int j = 0; j < len; j++){ if(q[x1 + j] == t[x2 + j]) x.mat++; else x.mis++; } x1 += len; x2 += len; break;
				case 1: x1 += len; x.ins += len; break;
				case 2: x2 += len; x.del += len; break;
			}
		}
		ora_cigar_concat(cigar, tmp_cigar->a, tmp_cigar->n);
		x.score += reg2->x.score;
		x.aln += reg2->x.aln; x.mat += reg2->x.mat; x.mis += reg2->x.mis; x.ins += reg2->x.ins; x.del += reg2->x.del;
		x.qe = reg2->x.qe; x.te = reg2->x.te;
		ora_cigar_concat(cigar, cigar_cache + reg2->cigar_off, reg2->cigar_len);
		reg1 = reg2;
	}
	if(x.te < len1 && x.qe < len2){
		w = ew;
		max_gap = ((ORA_MIN(len2 - x.qe, len1 - x.te) * M + x.score + (-T)) + (I < D ? D : I)) / (-E) + 1;
		if(max_gap < w) max_gap = w;
		for(;;){
			tmp_cigar->n = 0;
			y = ora_extend_shift(len2 - x.qe, pb2 + x.qe, len1 - x.te, pb1 + x.te, 1, x.score, -w, M, X, I, D, E, T, mem, tmp_cigar);
			if(y.qe == len2 - x.qe || y.te == len1 - x.te) break;
			if(x.te + y.te >= esti_regs[1]) break;
			if(w >= ew || w >= max_gap) break;
			w <<= 1;
		}
		x.score = y.score;
		x.aln += y.aln; x.mat += y.mat; x.mis += y.mis; x.ins += y.ins; x.del += y.del;
		x.qe += y.qe; x.te += y.te;
		ora_cigar_concat(cigar, tmp_cigar->a, tmp_cigar->n);
	}
	return x;
}

#endif
