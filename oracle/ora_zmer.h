/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see ora_util.h).
 *
 * ora_zmer.h — per-query z-mer table (A5) and query-vs-candidate z-mer matching (A6).
 * Restates:
 *   - index_single_read_seeds      reference hzm_aln.h:70-115
 *   - query_single_read_seeds      reference hzm_aln.h:173-224
 * The reference locates a z-mer through a 4^z bit-vector + rank (bitvec.h:289-343); only the
 * map  z-mer -> (first, count, dense index)  is observable, so the oracle keeps the distinct
 * retained z-mers in an ascending array and binary-searches it (dense index == position).
 */
#ifndef ORA_ZMER_H
#define ORA_ZMER_H

#include "ora_seq.h"

typedef struct { uint32_t mer, dir, off, len; } ora_zmer_t;                 /* hzm_t  */
typedef struct { uint32_t dir1, off1, dir2, off2, len1, len2, gid; } ora_zhit_t;   /* hzmp_t */
ORA_VEC(vec_zmer, ora_zmer_t)
ORA_VEC(vec_zhit, ora_zhit_t)

typedef struct {
	vec_zmer occ;        /* all canonical z-mers of the query sorted by (mer, off) */
	vec_u32  mers;       /* retained distinct z-mers (count < max_kcnt), ascending */
	vec_u32  first;      /* first occurrence in occ */
	vec_u32  cnt;
} ora_ztable_t;

static int ora_zmer_cmp(const void *pa, const void *pb){
	const ora_zmer_t *a = (const ora_zmer_t*)pa, *b = (const ora_zmer_t*)pb;
	if(a->mer != b->mer) return a->mer < b->mer ? -1 : 1;
	if(a->off != b->off) return a->off < b->off ? -1 : 1;
	return 0;
}

/* shared hp-compressed z-mer walk; BODY sees w_mer, w_dir, w_off, w_len */
#define ORA_ZMER_WALK(seq, slen, zsize, hz, BODY) do { \
	uint64_t _mask = 0xFFFFFFFFFFFFFFFFULL >> ((32 - (zsize)) << 1); \
	uint64_t _kmer = 0; uint32_t _i = 0, _j; uint8_t _b = 4; \
	zoff.n = 0; \
	for(_j = 0; _j < (slen); _j++){ \
		uint8_t _c = (seq)[_j]; \
		if((hz) && _c == _b) continue; \
		_b = _c; _i++; vec_u32_push(&zoff, _j); \
		_kmer = ((_kmer << 2) | _b) & _mask; \
		if(_i < (zsize)) continue; \
		uint64_t _rev = ora_revcomp_kmer(_kmer, (zsize)); \
		if(_rev == _kmer) continue; \
		uint32_t w_dir = _rev > _kmer ? 0u : 1u; \
		uint32_t w_mer = (uint32_t)(_rev > _kmer ? _kmer : _rev); \
		uint32_t w_off = zoff.a[_i - (zsize)]; \
		uint32_t w_len = (_j + 1 - w_off > 0xFFFFu) ? 0xFFFFu : _j + 1 - w_off; \
		BODY \
	} } while(0)

/* A5 (hzm_aln.h:70-115). The (mer,off) order is total, any correct sort is exact. */
static void ora_ztable_build(ora_ztable_t *zt, const uint8_t *seq, uint32_t slen, uint32_t zsize, int hz, uint32_t max_kcnt){
	vec_u32 zoff = {0};
	zt->occ.n = zt->mers.n = zt->first.n = zt->cnt.n = 0;
	ORA_ZMER_WALK(seq, slen, zsize, hz, {
		ora_zmer_t z; z.mer = w_mer; z.dir = w_dir; z.off = w_off; z.len = w_len;
		vec_zmer_push(&zt->occ, z);
	});
	qsort(zt->occ.a, zt->occ.n, sizeof(ora_zmer_t), ora_zmer_cmp);
	for(size_t i = 0; i < zt->occ.n; ){
		size_t j = i; while(j < zt->occ.n && zt->occ.a[j].mer == zt->occ.a[i].mer) j++;
		uint32_t c = (uint32_t)(j - i);
		/* hzm_aln.h:107: the running k-mer starts at 0 with count 0, so a real z-mer 0 is counted
		 * from its first element like any other; kept iff 0 < cnt < max_kcnt */
		if(c && c < max_kcnt){
			vec_u32_push(&zt->mers, zt->occ.a[i].mer); vec_u32_push(&zt->first, (uint32_t)i); vec_u32_push(&zt->cnt, c);
		}
		i = j;
	}
	vec_u32_free(&zoff);
}

static inline long ora_ztable_find(const ora_ztable_t *zt, uint32_t mer){
	size_t lo = 0, hi = zt->mers.n;
	while(lo < hi){ size_t mid = lo + (hi - lo) / 2; if(zt->mers.a[mid] < mer) lo = mid + 1; else hi = mid; }
	return (lo < zt->mers.n && zt->mers.a[lo] == mer) ? (long)lo : -1;
}

/* A6 (hzm_aln.h:173-224): candidate walked on its forward strand only */
static void ora_zmatch(const ora_ztable_t *zt, const uint8_t *cseq, uint32_t clen, uint32_t zsize, int hz, uint32_t max_kcnt, uint32_t max_var, vec_u8 *kcnts, vec_zhit *out){
	vec_u32 zoff = {0};
	out->n = 0;
	vec_u8_reserve(kcnts, zt->mers.n + 1);
	memset(kcnts->a, 0, zt->mers.n + 1);
	ORA_ZMER_WALK(cseq, clen, zsize, hz, {
		long idx = ora_ztable_find(zt, w_mer);
		if(idx < 0) continue;
		if(kcnts->a[idx] >= max_kcnt) continue;     /* u8 counter, hzm_aln.h:208-211 */
		kcnts->a[idx]++;
		for(uint32_t k = 0; k < zt->cnt.a[idx]; k++){
			const ora_zmer_t *p1 = &zt->occ.a[zt->first.a[idx] + k];
			uint32_t dv = p1->len > w_len ? p1->len - w_len : w_len - p1->len;
			if(dv > max_var) continue;
			ora_zhit_t h;
			h.dir1 = p1->dir; h.off1 = p1->off; h.dir2 = w_dir; h.len1 = p1->len; h.len2 = w_len; h.gid = 0;
			h.off2 = (p1->dir ^ w_dir) ? clen - (w_off + w_len) : w_off;
			vec_zhit_push(out, h);
		}
	});
	vec_u32_free(&zoff);
}

#endif
