/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see ora_util.h).
 *
 * ora_index.h — homopolymer-compressed k-mer index (A2) and per-query candidate search (A3).
 * Restates:
 *   - k-mer walk, canonical form, palindrome skip, Jenkins subsample
 *                               reference wtzmo.c:33-35, 249-285 (count), 286-318 (fill)
 *   - frequency cutoff / filter flags / per-k-mer (rd_id,dir) ordering
 *                               reference wtzmo.c:380-413, 337-345
 *   - candidate search          reference wtzmo.c:433-573 (incl. the two heap quirks at
 *                               528-531 and 563-567), hash hashset.h:452-462
 *
 * The reference keeps 1024 open-addressing hash sets; nothing observable depends on their
 * layout, so the oracle stores the index as a k-mer-sorted table and finds
 * k-mers by binary search.  The reference's k-way heap merge over (rd_id,dir,qoff) has a
 * total order (each query offset yields at most one cursor), so it is restated as a sort of
 * the expanded (rd_id,dir,qoff,len) tuples; duplicates are identical tuples.
 */
#ifndef ORA_INDEX_H
#define ORA_INDEX_H

#include "ora_seq.h"

#define ORA_KMER_MOD 1024u

static inline uint32_t ora_jenkins32(uint32_t key){   /* hashset.h:452-462 */
	key += (key << 12); key ^= (key >> 22);
	key += (key << 4);  key ^= (key >> 9);
	key += (key << 10); key ^= (key >> 2);
	key += (key << 7);  key ^= (key >> 12);
	return key;
}

typedef struct { uint64_t mer; uint32_t rd_dir; uint32_t pad; } ora_kocc_t;   /* rd_dir = rd_id<<1 | dir */
ORA_VEC(vec_kocc, ora_kocc_t)

typedef struct {
	uint64_t *mers;     /* distinct sampled k-mers, ascending */
	uint64_t *offs;     /* start of the k-mer's run in seeds[] */
	uint32_t *cnts;     /* run length (0 when filtered) */
	uint8_t  *flt;      /* 1: too frequent or singleton */
	size_t    n_mer;
	uint32_t *seeds;    /* rd_id<<1|dir, ascending inside each run */
	size_t    n_seed;
	uint32_t  max_kmer_freq;   /* resolved cutoff (wtzmo.c:380-393) */
	uint32_t  avg_rdlen;
} ora_kindex_t;

typedef struct {
	uint32_t ksize, hk, ksave, kovl, ncand;
	uint32_t max_kmer_freq;   /* -K, 0 = auto; updated in place like wt->max_kmer_freq */
} ora_kparams_t;

static int ora_kocc_cmp(const void *pa, const void *pb){
	const ora_kocc_t *a = (const ora_kocc_t*)pa, *b = (const ora_kocc_t*)pb;
	if(a->mer != b->mer) return a->mer < b->mer ? -1 : 1;
	if(a->rd_dir != b->rd_dir) return a->rd_dir < b->rd_dir ? -1 : 1;
	return 0;
}

/* enumerate sampled canonical hp-k-mers of one read; CB(mer, dir, qoff, qend_excl) */
#define ORA_KMER_WALK(st, rd, P, BODY) do { \
	uint64_t _mask = 0xFFFFFFFFFFFFFFFFULL >> ((32 - (P)->ksize) << 1); \
	uint64_t _kmer = 0, _off = (rd)->off; uint32_t _i = 0, _j, _len = (rd)->len; uint8_t _b = 4; \
	vec_u32 *_hz = &hzoff; _hz->n = 0; \
	for(_j = 0; _j < _len; _j++){ \
		uint8_t _c = (uint8_t)ora_base_at((st)->bits, _off + _j); \
		if((P)->hk && _c == _b) continue; \
		_b = _c; _i++; vec_u32_push(_hz, _j); \
		_kmer = ((_kmer << 2) | _b) & _mask; \
		if(_i < (P)->ksize) continue; \
		uint64_t _rev = ora_revcomp_kmer(_kmer, (P)->ksize); \
		if(_rev == _kmer) continue; \
		uint32_t w_dir = _rev > _kmer ? 0u : 1u; \
		uint64_t w_mer = _rev > _kmer ? _kmer : _rev; \
		uint32_t _kidx = ora_jenkins32((uint32_t)w_mer) % (ORA_KMER_MOD * (P)->ksave); \
		if(_kidx >= ORA_KMER_MOD) continue; \
		uint32_t w_qoff = _hz->a[_i - (P)->ksize]; uint32_t w_qend = _j + 1; \
		(void)w_dir; (void)w_qoff; (void)w_qend; \
		BODY \
	} } while(0)

static void ora_kindex_free(ora_kindex_t *ix){
	free(ix->mers); free(ix->offs); free(ix->cnts); free(ix->flt); free(ix->seeds);
	memset(ix, 0, sizeof(*ix));
}

/* A2: index reads [beg,end) (wtzmo.c:349-430) */
static void ora_kindex_build(ora_kindex_t *ix, const ora_store_t *st, uint32_t beg, uint32_t end, ora_kparams_t *P){
	vec_kocc occ = {0}; vec_u32 hzoff = {0};
	uint32_t n_rd = st->n_rd;
	uint64_t totlen = 0;
	ora_kindex_free(ix);
	if(n_rd){ for(uint32_t i = 0; i < n_rd; i++) totlen += st->reads.a[i].len; ix->avg_rdlen = (uint32_t)(totlen / n_rd); }
	else ix->avg_rdlen = 10000;
	for(uint32_t id = beg; id < end && id < n_rd; id++){
		const ora_read_t *rd = &st->reads.a[id];
		ORA_KMER_WALK(st, rd, P, {
			ora_kocc_t o; o.mer = w_mer; o.rd_dir = (id << 1) | w_dir; o.pad = 0;
			vec_kocc_push(&occ, o);
		});
	}
	qsort(occ.a, occ.n, sizeof(ora_kocc_t), ora_kocc_cmp);
	size_t n_mer = 0;
	for(size_t i = 0; i < occ.n; i++) if(i == 0 || occ.a[i].mer != occ.a[i-1].mer) n_mer++;
	ix->mers = (uint64_t*)ora_xrealloc(NULL, n_mer * 8);
	ix->offs = (uint64_t*)ora_xrealloc(NULL, n_mer * 8);
	ix->cnts = (uint32_t*)ora_xrealloc(NULL, n_mer * 4);
	ix->flt  = (uint8_t*) ora_xrealloc(NULL, n_mer);
	ix->n_mer = n_mer;
	/* counts saturate at 0xFFFF (wtzmo.c:276) */
	size_t m = 0; uint64_t ktot = 0;
	for(size_t i = 0; i < occ.n; ){
		size_t j = i; while(j < occ.n && occ.a[j].mer == occ.a[i].mer) j++;
		uint64_t c = j - i; if(c > 0xFFFFu) c = 0xFFFFu;
		ix->mers[m] = occ.a[i].mer; ix->cnts[m] = (uint32_t)c; ix->offs[m] = i; /* occ offset for now */
		ktot += c; m++; i = j;
	}
	if(P->max_kmer_freq < 2){      /* wtzmo.c:380-393 */
		uint32_t kavg = (uint32_t)(ktot / ((uint64_t)n_mer + 1));
		if(kavg < 20) kavg = 20;
		P->max_kmer_freq = kavg * 5;
	}
	ix->max_kmer_freq = P->max_kmer_freq;
	uint64_t off = 0;
	uint64_t *occ_off = (uint64_t*)ora_xrealloc(NULL, n_mer * 8);
	for(m = 0; m < n_mer; m++){    /* wtzmo.c:396-411 */
		uint32_t c = ix->cnts[m];
		occ_off[m] = ix->offs[m];
		ix->flt[m] = 0;
		if(c > P->max_kmer_freq){ c = 0; ix->flt[m] = 1; }
		ix->offs[m] = off; off += c;
		if(c <= 1) ix->flt[m] = 1;
		ix->cnts[m] = c;
	}
	ix->n_seed = off;
	ix->seeds = (uint32_t*)ora_xrealloc(NULL, off * 4);
	for(m = 0; m < n_mer; m++){    /* fill + per-k-mer sort (wtzmo.c:286-318, 337-345) */
		if(ix->flt[m]){ ix->cnts[m] = 0; continue; }
		for(uint32_t k = 0; k < ix->cnts[m]; k++) ix->seeds[ix->offs[m] + k] = occ.a[occ_off[m] + k].rd_dir;
	}
	free(occ_off);
	vec_kocc_free(&occ); vec_u32_free(&hzoff);
}

static inline long ora_kindex_find(const ora_kindex_t *ix, uint64_t mer){
	size_t lo = 0, hi = ix->n_mer;
	while(lo < hi){ size_t mid = lo + (hi - lo) / 2; if(ix->mers[mid] < mer) lo = mid + 1; else hi = mid; }
	return (lo < ix->n_mer && ix->mers[lo] == mer) ? (long)lo : -1;
}

typedef struct { uint32_t key, qoff, len; } ora_khit_t;   /* key = rd_id<<1|dir */
ORA_VEC(vec_khit, ora_khit_t)

static int ora_khit_cmp(const void *pa, const void *pb){
	const ora_khit_t *a = (const ora_khit_t*)pa, *b = (const ora_khit_t*)pb;
	if(a->key != b->key) return a->key < b->key ? -1 : 1;
	if(a->qoff != b->qoff) return a->qoff < b->qoff ? -1 : 1;
	return 0;
}

#define ORA_CAND_CMP(a, b) ((((a) & 0xFFFFFFFFu) > ((b) & 0xFFFFFFFFu)) ? 1 : ((((a) & 0xFFFFFFFFu) < ((b) & 0xFFFFFFFFu)) ? -1 : 0))
ORA_DEFINE_HEAP(ora_candheap, uint64_t, ORA_CAND_CMP)

#define ORA_CAND_NONE 0xFFFFFFFF00000000ULL

/* The (id,dir,ol) groups of one query in merge order: the pure part of A3 (what a GPU computes). */
typedef struct { uint32_t key, ol, cnt; } ora_kgroup_t;
ORA_VEC(vec_kgroup, ora_kgroup_t)

static void ora_query_groups(const ora_store_t *st, const ora_kindex_t *ix, const ora_kparams_t *P, uint32_t pbid, vec_kgroup *groups, vec_khit *hits){
	vec_u32 hzoff = {0};
	const ora_read_t *rd = &st->reads.a[pbid];
	uint32_t pblen = rd->len;
	uint32_t pblen_up = (uint32_t)(pblen * 1.2);     /* wtzmo.c:445 (double) */
	hits->n = 0; groups->n = 0;
	ORA_KMER_WALK(st, rd, P, {
		long h = ora_kindex_find(ix, w_mer);
		if(h < 0) continue;
		if(ix->flt[h]) continue;
		uint32_t len = w_qend - w_qoff; if(len > 0xFFFFu) len = 0xFFFFu;
		for(uint32_t k = 0; k < ix->cnts[h]; k++){
			uint32_t s = ix->seeds[ix->offs[h] + k];
			if((s >> 1) == pbid) continue;
			if(st->reads.a[s >> 1].len > pblen_up) continue;
			ora_khit_t t; t.key = s; t.qoff = w_qoff; t.len = len;
			vec_khit_push(hits, t);
		}
	});
	qsort(hits->a, hits->n, sizeof(ora_khit_t), ora_khit_cmp);
	for(size_t i = 0; i < hits->n; ){
		uint32_t ol = 0, lst = 0, cnt = 0; size_t j = i;
		for(; j < hits->n && hits->a[j].key == hits->a[i].key; j++){   /* wtzmo.c:558-561 */
			const ora_khit_t *t = &hits->a[j];
			if(t->qoff >= lst) ol += t->len; else ol += t->qoff + t->len - lst;
			lst = t->qoff + t->len; cnt++;
		}
		ora_kgroup_t g; g.key = hits->a[i].key; g.ol = ol; g.cnt = cnt;
		vec_kgroup_push(groups, g);
		i = j;
	}
	vec_u32_free(&hzoff);
}

/* The order-sensitive tail of A3 (wtzmo.c:516-571): x1/x2 strand merge + top-ncand heap. */
static void ora_candidates_from_groups(const ora_kgroup_t *g, size_t ng, uint32_t kovl, uint32_t ncand, vec_u64 *cand){
	uint64_t x1 = ORA_CAND_NONE, x2;
	vec_u64_reserve(cand, cand->n + ng + 2);
	for(size_t i = 0; i < ng; i++){
		uint32_t ol = g[i].ol;
		if(ol < kovl) continue;
		x2 = (((uint64_t)(g[i].key >> 1)) << 32) | ol;
		if((x1 >> 32) == (x2 >> 32)){ x1 = (x1 & 0xFFFFFFFFu) > (x2 & 0xFFFFFFFFu) ? x1 : x2; }
		else if(x1 == ORA_CAND_NONE){ x1 = x2; }
		else {
			if(cand->n >= ncand){
				if((cand->a[0] & 0xFFFFFFFFu) < ol) ora_candheap_replace_top(cand->a, cand->n, x1, NULL);   /* quirk 528-531 */
			} else ora_candheap_push(cand->a, &cand->n, x1, NULL);
			x1 = x2;
		}
	}
	/* final flush compares against ol == 0 (wtzmo.c:563-571): dropped when the heap is full */
	if(cand->n >= ncand){ /* (top & M) < 0 is never true */ }
	else { vec_u64_reserve(cand, cand->n + 1); ora_candheap_push(cand->a, &cand->n, x1, NULL); }
}

#define ORA_CAND_GT(a, b) (((b) & 0xFFFFFFFFu) > ((a) & 0xFFFFFFFFu))
ORA_DEFINE_SORT(ora_sort_cand_desc, uint64_t, ORA_CAND_GT)

#endif
