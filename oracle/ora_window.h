/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see ora_util.h).
 *
 * ora_window.h — zmo engine, window detection and window chaining (A7z).
 * Restates:
 *   - process_hzmps                       reference hzm_aln.h:1184-1186
 *   - calculate_median_value              reference hzm_aln.h:316-343
 *   - potential_paired_kmers_windows      reference hzm_aln.h:410-578 (fast-chaining branch;
 *                                         HZM_FAST_WINDOW_KMER_CHAINING is forced to 1 at wtzmo.c:1540)
 *   - merge_paired_kmers_window           reference hzm_aln.h:580-656
 *   - chaining_wtseedv                    reference hzm_aln.h:658-713
 * Integer widths follow the C promotions of the reference's bit-fields: 31-bit offsets and
 * 16-bit lengths promote to int, mix with uint32_t operands as unsigned 32-bit.
 */
#ifndef ORA_WINDOW_H
#define ORA_WINDOW_H

#include "ora_zmer.h"

typedef struct {
	uint32_t pb2;
	uint32_t ovl;        /* :29 */
	uint32_t dir;        /* :1  */
	uint32_t closed;     /* :2  */
	int32_t  beg[2], end[2];
	uint32_t anchors[2];
} ora_win_t;                                             /* wt_seed_t, hzm_aln.h:62-67 */
ORA_VEC(vec_win, ora_win_t)

#define ORA_OVL29(x) ((uint32_t)(x) & 0x1FFFFFFFu)
#define ORA_KWIN_MAX_OFFSET_DEV 50

#define ORA_ZHIT_KEY12(h) ((((int64_t)(h).off1) << 32) | (int64_t)(h).off2)
#define ORA_ZHIT_GT12(a, b) (ORA_ZHIT_KEY12(a) > ORA_ZHIT_KEY12(b))
ORA_DEFINE_SORT(ora_sort_zhit_off12, ora_zhit_t, ORA_ZHIT_GT12)

#define ORA_ZHIT_GT1(a, b) ((a).off1 > (b).off1)
ORA_DEFINE_SORT(ora_sort_zhit_off1, ora_zhit_t, ORA_ZHIT_GT1)

/* index sort by off2 of the referenced element: ctx = const ora_zhit_t* */
#define ORA_IDX_GT_OFF2(a, b) (((const ora_zhit_t*)ctx)[a].off2 > ((const ora_zhit_t*)ctx)[b].off2)
ORA_DEFINE_SORT(ora_sort_idx_by_off2, uint32_t, ORA_IDX_GT_OFF2)

/* hzm_aln.h:316-343 — quick-select; permutes rs */
static int32_t ora_median(int32_t *rs, int32_t size){
	int32_t i, j, key, mid, beg, end, tmp;
	if(size == 0) return 0;
	beg = 0; end = size - 1;
	while(beg < end){
		mid = beg + (end - beg) / 2;
		if(rs[beg] > rs[mid]){ tmp = rs[beg]; rs[beg] = rs[mid]; rs[mid] = tmp; }
		if(rs[mid] > rs[end]){
			tmp = rs[end]; rs[end] = rs[mid]; rs[mid] = tmp;
			if(rs[beg] > rs[mid]){ tmp = rs[beg]; rs[beg] = rs[mid]; rs[mid] = tmp; }
		}
		key = rs[mid];
		i = beg + 1; j = end - 1;
		for(;;){
			while(key > rs[i]) i++;
			while(rs[j] > key) j--;
			if(i < j){ tmp = rs[i]; rs[i] = rs[j]; rs[j] = tmp; i++; j--; }
			else break;
		}
		if(i == j){ i++; j--; }
		if(i <= size / 2) beg = i; else end = j;
	}
	return rs[size / 2];
}

typedef struct { vec_u32 ts; vec_i32 as; vec_u32 wb, we, wo; } ora_winscratch_t;

/* hzm_aln.h:410-578 */
static uint32_t ora_scan_windows(const ora_zhit_t *rs, int dir, uint32_t beg, uint32_t end, int bound,
		vec_win *wins, vec_zhit *anchors, ora_winscratch_t *sc, uint32_t zsize, uint32_t kwin, uint32_t zovl){
	uint32_t i, j, n = 0, n2, ol, ol2, s, t, lst, ret;
	while(beg < end){
		const ora_zhit_t *p = &rs[beg];
		if((p->dir1 ^ p->dir2 ^ (uint32_t)dir) || (int)p->off1 < bound) beg++;
		else break;
	}
	for(i = beg; i < end; i++){ if(rs[i].dir1 ^ rs[i].dir2 ^ (uint32_t)dir) continue; n++; }
	if(n * zsize < zovl) return 0;
	sc->ts.n = 0; vec_u32_reserve(&sc->ts, n + 1);
	for(i = beg; i < end; i++){ if(rs[i].dir1 ^ rs[i].dir2 ^ (uint32_t)dir) continue; sc->ts.a[sc->ts.n++] = i; }
	uint32_t *ts = sc->ts.a;
	ora_sort_idx_by_off2(ts, n, (void*)rs);
	sc->wb.n = sc->we.n = sc->wo.n = 0;
	vec_u32_reserve(&sc->wb, n + 1); vec_u32_reserve(&sc->we, n + 1); vec_u32_reserve(&sc->wo, n + 1);
	ol = 0; lst = 0; n2 = 0;
	for(i = j = 0; i < n; i++){
		const ora_zhit_t *p = &rs[ts[i]];
		while((uint32_t)p->off2 + p->len2 > rs[ts[j]].off2 + kwin){
			const ora_zhit_t *p0 = &rs[ts[j++]];
			const ora_zhit_t *p1 = &rs[ts[j]];
			s = p1->off2; t = p0->off2 + p0->len2;
			ol2 = s < t ? t - s : 0;
			ol = ol + ol2 - p0->len2;
		}
		ol += (p->off2 > lst) ? p->len2 : p->off2 + p->len2 - lst;
		lst = p->off2 + p->len2;
		if(ol >= zovl){
			if(n2 && ( rs[ts[i]].off2 <= rs[ts[sc->we.a[n2-1]]].off2 + kwin / 3 ||
			           rs[ts[j]].off2 <= rs[ts[sc->wb.a[n2-1]]].off2 + kwin / 3 )){
				if(ol > sc->wo.a[n2-1]){ sc->wb.a[n2-1] = j; sc->we.a[n2-1] = i; sc->wo.a[n2-1] = ol; }
			} else { sc->wb.a[n2] = j; sc->we.a[n2] = i; sc->wo.a[n2] = ol; n2++; }
		}
	}
	ret = 0;
	for(i = 0; i < n2; i++){
		size_t size = anchors->n;
		int offset, off;
		sc->as.n = 0;
		for(j = sc->wb.a[i]; j <= sc->we.a[i]; j++){
			const ora_zhit_t *p = &rs[ts[j]];
			vec_i32_push(&sc->as, (int)p->off1 - (int)p->off2);
		}
		offset = ora_median(sc->as.a, (int32_t)sc->as.n);
		ol = lst = 0;
		for(j = sc->wb.a[i]; j <= sc->we.a[i]; j++){
			const ora_zhit_t *p = &rs[ts[j]];
			off = (int)p->off1 - (int)p->off2;
			if(off < offset - ORA_KWIN_MAX_OFFSET_DEV || off > offset + ORA_KWIN_MAX_OFFSET_DEV) continue;
			vec_zhit_push(anchors, *p);
			ol += (p->off2 > lst) ? p->len2 : p->off2 + p->len2 - lst;
			lst = p->off2 + p->len2;
		}
		if(anchors->n == size) continue;
		ora_sort_zhit_off1(anchors->a + size, anchors->n - size, NULL);
		ora_win_t *w = vec_win_next(wins);
		w->pb2 = 0;
		w->closed = 0; w->dir = (uint32_t)dir;
		w->anchors[0] = (uint32_t)size; w->anchors[1] = 0;
		w->beg[0] = w->beg[1] = 0x7FFFFFFF; w->end[0] = w->end[1] = 0;
		w->ovl = ORA_OVL29(ol);
		ol = lst = 0;
		for(size_t k = size; k < anchors->n; k++){
			const ora_zhit_t *p = &anchors->a[k];
			ol += (p->off1 > lst) ? p->len1 : p->off1 + p->len1 - lst;
			lst = p->off1 + p->len1;
			if((int)p->off1 < w->beg[0]) w->beg[0] = (int)p->off1;
			if((int)(p->off1 + p->len1) > w->end[0]) w->end[0] = (int)(p->off1 + p->len1);
			if((int)p->off2 < w->beg[1]) w->beg[1] = (int)p->off2;
			if((int)(p->off2 + p->len2) > w->end[1]) w->end[1] = (int)(p->off2 + p->len2);
		}
		if(ol * 2 < zovl){
			anchors->n = size; wins->n--;
		} else if(ret && (w->end[1] <= (int)((uint32_t)wins->a[wins->n - 2].end[1] + kwin / 3) && ol <= wins->a[wins->n - 2].ovl)){
			anchors->n = size; wins->n--;
		} else {
			ret++;
			w->ovl = ORA_OVL29(ol);
			w->anchors[1] = (uint32_t)anchors->n;
		}
	}
	return ret;
}

/* hzm_aln.h:580-656. rs must have one readable element past n (the reference reads it too). */
static uint32_t ora_merge_windows(const ora_zhit_t *rs, uint32_t n_rs, int dir, vec_win *wins, vec_zhit *anchors,
		ora_winscratch_t *sc, uint32_t zsize, uint32_t kwin, uint32_t kstep, uint32_t zovl){
	ora_zhit_t P; memset(&P, 0, sizeof(P)); P.off1 = 0x1FFFFFu; P.len1 = 0x3FFu;
	const ora_zhit_t *p, *p0, *p1;
	uint32_t i, j, n, a, ol, ol2, lst, wlst, s, t, ret;
	int nxt;
	ol = 0; lst = 0; wlst = 0; ret = 0;
	for(j = 0; j < n_rs; j++){ if(rs[j].dir1 ^ rs[j].dir2 ^ (uint32_t)dir) continue; break; }
	if(j == n_rs) return 0;
	p0 = &rs[j];
	p = p0;
	for(i = j; i <= n_rs; i++){
		if(i < n_rs){ p = &rs[i]; if(p->dir1 ^ p->dir2 ^ (uint32_t)dir) continue; }
		else p = &P;
		if((uint32_t)p->off1 > (uint32_t)p0->off1 + kwin){
			if(ol >= zovl){
				if((n = ora_scan_windows(rs, dir, j, i, (int)wlst, wins, anchors, sc, zsize, kwin, zovl))){
					for(a = 0; a < n; a++){
						int e0 = wins->a[wins->n + a - n].end[0] + 20;
						if((int)wlst < e0) wlst = (uint32_t)e0;
					}
					ret += n;
					p0 = p; ol = p->len1; lst = p->off1 + p->len1; j = i;
				} else {
					nxt = (int)(p0->off1 + kstep);
					while((int)p0->off1 < nxt && j < i){
						p1 = &rs[++j];
						s = ORA_MAX(p0->off1, p1->off1);
						t = ORA_MIN(p0->off1 + p0->len1, p1->off1 + p1->len1);
						ol2 = s < t ? t - s : 0;
						ol = ol + ol2 - p0->len1;
						p0 = p1;
					}
				}
			}
			if(p->off1 == P.off1) break;
			while((uint32_t)p->off1 > (uint32_t)p0->off1 + kwin){
				p1 = &rs[++j];
				s = ORA_MAX(p0->off1, p1->off1);
				t = ORA_MIN(p0->off1 + p0->len1, p1->off1 + p1->len1);
				ol2 = s < t ? t - s : 0;
				ol = ol + ol2 - p0->len1;
				p0 = p1;
			}
		} else {
			if(p->off1 >= lst) ol += p->len1;
			else if((int)(p->off1 + p->len1) > (int)lst) ol += p->off1 + p->len1 - lst;
			else continue;
			lst = p->off1 + p->len1;
		}
	}
	return ret;
}

/* hzm_aln.h:658-713 — colinear chain of windows; marks members closed=0, others closed=1 */
static int ora_chain_windows(ora_win_t *regs, uint32_t beg, uint32_t end, int W, vec_i32 *mem){
	const int max_overhang = 0;
	const float band_penalty = 0.05f;
	uint32_t i, j; int mw, bt, band;
	uint32_t n = end - beg;
	mem->n = 0; vec_i32_reserve(mem, 2 * (size_t)n + 2);
	int32_t *weight = mem->a, *back = mem->a + n;
	for(i = 0; i < n; i++){ weight[i] = 0; back[i] = -1; }
	mw = -1000000; bt = -1;
	for(i = beg; i < end; i++){
		ora_win_t *r1 = &regs[i];
		r1->closed = 1;
		weight[i - beg] += (int32_t)r1->ovl;
		if(weight[i - beg] > mw){ mw = weight[i - beg]; bt = (int)i; }
		for(j = i + 1; j < end; j++){
			const ora_win_t *r2 = &regs[j];
			if(r2->beg[1] + max_overhang < r1->end[1]) continue;
			if(r2->beg[0] + max_overhang < r1->end[0]) continue;
			if(r2->beg[0] - r1->end[0] > W && r2->beg[1] - r1->end[1] > W) break;
			band = ORA_ABSDIFF(r2->beg[0] - r1->end[0], r2->beg[1] - r1->end[1]);
			if(band > W) continue;
			band = (int)(band * band_penalty);
			if(weight[j - beg] < weight[i - beg] - band){ weight[j - beg] = weight[i - beg] - band; back[j - beg] = (int)i; }
		}
	}
	mw = 0;
	while(bt >= 0){
		ora_win_t *r1 = &regs[bt];
		r1->closed = 0;
		mw += r1->end[0] - r1->beg[0];
		bt = back[bt - (int)beg];
	}
	return mw;
}

#endif
