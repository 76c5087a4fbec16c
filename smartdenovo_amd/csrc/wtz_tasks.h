/*
 * wtz_tasks.h — the per-pair and per-window tasks launched by the stage functions of the C-ABI.
 *
 *   wtz_task_pair      one (query, candidate): A6 z-mer matches -> gate (wtzmo.c:857) ->
 *                      zmo: exact (off1,off2) order + windows + chain per strand (wtzmo.c:887-914)
 *                      dmo: dot-matrix alignment (wtzmo.c:858-886)
 *   wtz_task_winalign  one chain window: seed-anchored alignment A9 (wtzmo.c:1019-1028)
 *   wtz_task_stitch    one (pair, strand): stitch windows A10 (wtzmo.c:1030, hzm_aln.h:1345-1486)
 */
#ifndef WTZ_TASKS_H
#define WTZ_TASKS_H

#include "wtz_sw.h"
#include "wtz_dotmatrix.h"

typedef struct {
	wtz_reads_t R; wtz_zindex_t Z; const wtz_params_t *P; wtz_pool_t *pool;
} wtz_env_t;

WTZ_HD void wtz_task_pair(uint32_t t, const wtz_env_t &V, const uint32_t *qid, const uint32_t *cid, wtz_pairres_t *res){
	const wtz_params_t *P = V.P;
	const uint32_t q = qid[t], c = cid[t];
	wtz_pairres_t r; memset(&r, 0, sizeof r);
	wtz_vec<wtz_zhit_t> cache; cache.init(V.pool, 0);
	{   /* size the match list from the candidate's z-mer count to avoid regrowth in the common case */
		uint32_t cn = (uint32_t)(V.Z.zoff[c + 1] - V.Z.zoff[c]);
		cache.reserve(cn / 2 + 64);
	}
	if(!wtz_zmatch(V.Z, q, c, V.R.rdlen[c], P->max_kmer_var, cache)){ r.bad = 1; res[t] = r; return; }
	r.n_hits = cache.n;
	if(cache.n * P->zsize < P->ztot){ r.gate = 0; res[t] = r; return; }
	r.gate = 1;
	if(P->dot_matrix){
		wtz_dm_result_t d = wtz_dot_matrix_align(cache, V.pool, (int32_t)V.R.rdlen[q], (int32_t)V.R.rdlen[c], P, &r.bad);
		r.dm_score = d.score; r.dm_qb = d.qb; r.dm_qe = d.qe; r.dm_tb = d.tb; r.dm_te = d.te; r.dm_dir = d.dir;
		res[t] = r; return;
	}
	if(!cache.reserve(cache.n + 1)){ r.bad = 1; res[t] = r; return; }
	memset(&cache.a[cache.n], 0, sizeof(wtz_zhit_t));            /* the element the reference reads past the end */
	wtz_sort_exact(cache.a, (size_t)cache.n, wtz_gt_off12());       /* process_hzmps, hzm_aln.h:1184-1186 */
	const uint32_t n = cache.n;
	wtz_winscratch_t sc;
	sc.ts = (uint32_t*)wtz_pool_alloc(V.pool, (size_t)(n + 1) * 4 * 5);
	if(sc.ts == NULL){ r.bad = 1; res[t] = r; return; }
	sc.as = (int32_t*)(sc.ts + (n + 1)); sc.wb = sc.ts + 2 * (n + 1); sc.we = sc.ts + 3 * (n + 1); sc.wo = sc.ts + 4 * (n + 1);
	for(uint32_t dir = 0; dir < 2; dir++){
		wtz_vec<wtz_win_t> wins; wins.init(V.pool, 16);
		wtz_vec<wtz_zhit_t> anchors; anchors.init(V.pool, n + 16);
		if(wtz_merge_windows(cache.a, n, dir, wins, anchors, sc, P->zsize, P->kwin, P->kstep, P->zovl) == 0){
			if(wins.bad || anchors.bad) r.bad = 1;
			continue;
		}
		if(wins.bad || anchors.bad){ r.bad = 1; continue; }
		int32_t *mem = (int32_t*)wtz_pool_alloc(V.pool, (size_t)wins.n * 8 + 8);
		if(mem == NULL){ r.bad = 1; continue; }
		r.ovl[dir] = WTZ_OVL29(wtz_chain_windows(wins.a, wins.n, P->W, mem));
		if(r.ovl[dir] < P->ztot) continue;
		/* keep the chain members in place (compact to the front); their anchors[] keep indexing `anchors` */
		uint32_t k = 0;
		for(uint32_t j = 0; j < wins.n; j++){ if(wins.a[j].closed) continue; wins.a[k++] = wins.a[j]; }
		r.nwin[dir] = k; r.win[dir] = wins.a; r.anchors[dir] = anchors.a; r.nanchors[dir] = anchors.n;
	}
	res[t] = r;
}

/* ---------------- alignment ---------------- */
typedef struct {                 /* one chain window to align */
	uint32_t item;               /* index of the (pair,strand) item it belongs to */
	uint32_t widx;               /* window index inside the pair's chain */
} wtz_wintask_t;

typedef struct { wtz_aln_t x; uint32_t *cigar; uint32_t cigar_len; uint32_t pass; } wtz_reg_t;   /* aln_reg_t + pass flag (wtzmo.c:1026) */

typedef struct {                 /* one (pair,strand) to align */
	uint32_t q, c, dir;
	const wtz_win_t *win; const wtz_zhit_t *anchors; uint32_t nwin;
	wtz_reg_t *regs;             /* nwin entries, filled by wtz_task_winalign */
} wtz_alnitem_t;

typedef struct {
	wtz_aln_t x; uint32_t n_regs; uint32_t cigar_len; uint32_t *cigar; int32_t bad;
	unsigned long long cells_shift, cells_fixed, cells_global;
} wtz_alnres_dev_t;

WTZ_HD wtz_readview wtz_view(const wtz_reads_t &R, uint32_t id, uint32_t rev){ wtz_readview v; v.bits = R.bits; v.off = R.rdoff[id]; v.len = R.rdlen[id]; v.rev = rev; return v; }

WTZ_HD void wtz_task_winalign(uint32_t t, const wtz_env_t &V, const wtz_wintask_t *tasks, const wtz_alnitem_t *items){
	const wtz_params_t *P = V.P;
	const wtz_alnitem_t &it = items[tasks[t].item];
	const wtz_win_t &w = it.win[tasks[t].widx];
	wtz_reg_t reg; memset(&reg, 0, sizeof reg);
	wtz_cigar_t cigar, tmp; cigar.init(V.pool, 64); tmp.init(V.pool, 64);
	wtz_swmem_t mem; wtz_swmem_init(mem, V.pool);
	reg.x = wtz_align_window(wtz_view(V.R, it.q, 0), wtz_view(V.R, it.c, it.dir), w, it.anchors, cigar, mem, tmp, P);
	reg.cigar = cigar.a; reg.cigar_len = cigar.n;
	reg.pass = !(reg.x.aln * 2 < (int32_t)P->zovl || (float)reg.x.mat < (float)reg.x.aln * P->min_id);
	if(cigar.bad || tmp.bad || mem.bad) reg.pass = 2;        /* pool exhausted */
	it.regs[tasks[t].widx] = reg;
}

/* A10 for one item; hzm_aln.h:1345-1486 with esti_regs = {0, len1} (wtzmo.c:1030) */
WTZ_HD void wtz_task_stitch(uint32_t t, const wtz_env_t &V, const wtz_alnitem_t *items, wtz_alnres_dev_t *out){
	const wtz_params_t *P = V.P;
	const wtz_alnitem_t &it = items[t];
	const int32_t M = P->M, X = P->X, I = P->O, D = P->O, E = P->E, T = P->T, ew = P->ew;
	wtz_alnres_dev_t r; memset(&r, 0, sizeof r);
	const wtz_readview pb1 = wtz_view(V.R, it.q, 0), pb2 = wtz_view(V.R, it.c, it.dir);
	const int32_t len1 = (int32_t)pb1.len, len2 = (int32_t)pb2.len;
	const int32_t esti0 = 0, esti1 = len1;
	uint32_t first = 0xFFFFFFFFu, nreg = 0;
	for(uint32_t k = 0; k < it.nwin; k++){ if(it.regs[k].pass == 2) r.bad = 1; if(it.regs[k].pass == 1){ if(first == 0xFFFFFFFFu) first = k; nreg++; } }
	r.n_regs = nreg;
	if(nreg == 0 || r.bad){ out[t] = r; return; }
	wtz_cigar_t cigar, tmp; cigar.init(V.pool, 256); tmp.init(V.pool, 64);
	wtz_swmem_t mem; wtz_swmem_init(mem, V.pool);
	wtz_aln_t x = it.regs[first].x, y; memset(&y, 0, sizeof y);
	const int32_t init_score = 100 * M;
	int32_t w, max_gap, score;
	if(x.qb && x.tb){
		w = ew;
		max_gap = ((WTZ_MIN(x.qb, x.tb) * M + x.score + init_score + (-T)) + (I < D ? D : I)) / (-E) + 1;
		if(max_gap < w) max_gap = w;
		for(;;){
			tmp.n = 0;
			y = wtz_extend_shift(x.qb, pb2.sub(x.qb - 1, -1), x.tb, pb1.sub(x.tb - 1, -1), x.score + init_score, -w, M, X, I, D, E, T, mem, tmp, &r.cells_shift);
			if(y.qe == x.qb || y.te == x.tb) break;
			if(x.tb - y.te <= esti0) break;
			if(w >= ew || w >= max_gap) break;
			w <<= 1;
		}
		x.score = y.score - init_score;
		x.aln += y.aln; x.mat += y.mat; x.mis += y.mis; x.ins += y.ins; x.del += y.del;
		x.qb -= y.qe; x.tb -= y.te;
		wtz_cigar_reverse(tmp.a, tmp.n);
		wtz_cigar_concat(cigar, tmp.a, tmp.n);
	}
	wtz_cigar_concat(cigar, it.regs[first].cigar, it.regs[first].cigar_len);
	const wtz_reg_t *reg1 = &it.regs[first];
	for(uint32_t k = first + 1; k < it.nwin; k++){
		if(it.regs[k].pass != 1) continue;
		const wtz_reg_t *reg2 = &it.regs[k];
		const int32_t dq = reg2->x.qb - reg1->x.qe, dt = reg2->x.tb - reg1->x.te;
		const wtz_seq_packed q = pb2.sub(reg1->x.qe, 1), tt = pb1.sub(reg1->x.te, 1);
		w = P->w;
		for(;;){
			if(w < WTZ_ABSDIFF(dq, dt)){ w <<= 1; continue; }
			score = wtz_global_banded(dq, q, dt, tt, M, X, -I, -E, -D, -E, w, mem, tmp);
			if(score < 0 && w < P->W && w < WTZ_MAX(dq, dt)) w <<= 1;
			else break;
		}
		x.score += score;
		x.qe = reg2->x.qb; x.te = reg2->x.tb;
		int32_t x1 = 0, x2 = 0;
		for(uint32_t idx = 0; idx < tmp.n; idx++){
			int32_t op = (int32_t)(tmp.a[idx] & 0xF), len = (int32_t)(tmp.a[idx] >> 4);
			x.aln += len;
			if(op == 0){ for(int32_t j = 0; j < len; j++){ if(q.at(x1 + j) == tt.at(x2 + j)) x.mat++; else x.mis++; } x1 += len; x2 += len; }
			else if(op == 1){ x1 += len; x.ins += len; }
			else if(op == 2){ x2 += len; x.del += len; }
		}
		wtz_cigar_concat(cigar, tmp.a, tmp.n);
		x.score += reg2->x.score;
		x.aln += reg2->x.aln; x.mat += reg2->x.mat; x.mis += reg2->x.mis; x.ins += reg2->x.ins; x.del += reg2->x.del;
		x.qe = reg2->x.qe; x.te = reg2->x.te;
		wtz_cigar_concat(cigar, reg2->cigar, reg2->cigar_len);
		reg1 = reg2;
	}
	if(x.te < len1 && x.qe < len2){
		w = ew;
		max_gap = ((WTZ_MIN(len2 - x.qe, len1 - x.te) * M + x.score + (-T)) + (I < D ? D : I)) / (-E) + 1;
		if(max_gap < w) max_gap = w;
		for(;;){
			tmp.n = 0;
			y = wtz_extend_shift(len2 - x.qe, pb2.sub(x.qe, 1), len1 - x.te, pb1.sub(x.te, 1), x.score, -w, M, X, I, D, E, T, mem, tmp, &r.cells_shift);
			if(y.qe == len2 - x.qe || y.te == len1 - x.te) break;
			if(x.te + y.te >= esti1) break;
			if(w >= ew || w >= max_gap) break;
			w <<= 1;
		}
		x.score = y.score;
		x.aln += y.aln; x.mat += y.mat; x.mis += y.mis; x.ins += y.ins; x.del += y.del;
		x.qe += y.qe; x.te += y.te;
		wtz_cigar_concat(cigar, tmp.a, tmp.n);
	}
	r.x = x; r.cigar = cigar.a; r.cigar_len = cigar.n;
	if(cigar.bad || tmp.bad || mem.bad) r.bad = 1;
	out[t] = r;
}

#endif
