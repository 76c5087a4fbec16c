/*
 * wtz_seed.h — task bodies for the index / seed-lookup half of the path.
 *
 *   K-idx  : hp-compressed canonical k-mer scan of every read + subsample (A2: wtzmo.c:249-318)
 *   K-zidx : per-read z-mer lists and sorted tables, built once for all reads
 *            (A5: hzm_aln.h:70-115, and the candidate-side walk of A6: hzm_aln.h:185-211)
 *   K-seed : per-query seed lookup -> (read,strand) groups with union length `ol`
 *            -> candidate heap (A3: wtzmo.c:433-573)
 *
 * MI355X layout notes.  The k-mer table is an open-addressing hash of 16-byte {k-mer, off<<16|cnt}
 * slots holding only k-mers that pass the frequency filter (2 <= cnt <= K): singletons and
 * over-represented k-mers behave exactly like absent k-mers on the query path (wtzmo.c:473-478),
 * and at PacBio error rates >90% of distinct k-mers are singletons, so the table of an E. coli
 * run is a few tens of MB — resident in the 256 MiB Infinity Cache.  One probe = one 16-byte load.
 * The reference's k-way heap merge over (rd_id,dir,qoff) is replaced by an order-free per-query
 * accumulator keyed by (rd_id,dir): tuples arrive in qoff order, which is all the union-length
 * recurrence (wtzmo.c:558-560) needs; only the distinct keys are sorted afterwards.
 */
#ifndef WTZ_SEED_H
#define WTZ_SEED_H

#include "wtz_common.h"

#define WTZ_KMER_MOD 1024u
#define WTZ_KEMPTY 0xFFFFFFFFFFFFFFFFull

typedef struct { uint64_t key, val; } wtz_kslot_t;     /* val = off<<16 | cnt */

typedef struct {
	const uint64_t *bits; const uint64_t *rdoff; const uint32_t *rdlen; uint32_t n_reads;
} wtz_reads_t;

/* ---- shared hp-compressed k-mer walk. F(mer, dir, qoff, qend) is called for every sampled k-mer ---- */
template<typename F>
WTZ_HD void wtz_kmer_walk(const wtz_reads_t &R, uint32_t rid, uint32_t ksize, uint32_t hk, uint32_t ksave, F &f){
	const uint64_t mask = 0xFFFFFFFFFFFFFFFFULL >> ((32 - ksize) << 1);
	const uint64_t off = R.rdoff[rid]; const uint32_t len = R.rdlen[rid];
	uint32_t ring[32];                      /* start positions of the last ksize hp-runs (hzoff, wtzmo.c:461) */
	uint64_t kmer = 0; uint32_t i = 0; uint32_t b = 4;
	uint64_t word = 0;
	for(uint32_t j = 0; j < len; j++){
		uint64_t p = off + j;
		if(j == 0 || (p & 31u) == 0) word = R.bits[p >> 5];
		uint32_t c = (uint32_t)((word >> (((~p) & 31u) << 1)) & 3u);
		if(hk && c == b) continue;
		b = c; i++;
		ring[(i - 1) & 31u] = j;
		kmer = ((kmer << 2) | b) & mask;
		if(i < ksize) continue;
		uint64_t rev = wtz_revcomp_kmer(kmer, ksize);
		if(rev == kmer) continue;
		uint32_t dir = rev > kmer ? 0u : 1u;
		uint64_t mer = rev > kmer ? kmer : rev;
		uint32_t kidx = wtz_jenkins32((uint32_t)mer) % (WTZ_KMER_MOD * ksave);
		if(kidx >= WTZ_KMER_MOD) continue;
		f(mer, dir, ring[(i - ksize) & 31u], j + 1);
	}
}

/* ================= K-idx ================= */
struct wtz_kcount_f { uint32_t n; WTZ_HDM void operator()(uint64_t, uint32_t, uint32_t, uint32_t){ n++; } };
struct wtz_kfill_f  { uint64_t *keys; uint32_t *vals; uint64_t pos; uint32_t rid;
	WTZ_HDM void operator()(uint64_t mer, uint32_t dir, uint32_t, uint32_t){ keys[pos] = mer; vals[pos] = (rid << 1) | dir; pos++; } };

/* task: count sampled k-mers of read id_beg + t */
WTZ_HD void wtz_task_kcount(uint32_t t, wtz_reads_t R, uint32_t id_beg, uint32_t ksize, uint32_t hk, uint32_t ksave, uint64_t *cnt){
	wtz_kcount_f f; f.n = 0;
	wtz_kmer_walk(R, id_beg + t, ksize, hk, ksave, f);
	cnt[t] = f.n;
}
/* task: write (k-mer, rd<<1|dir) of read id_beg + t at offs[t] */
WTZ_HD void wtz_task_kfill(uint32_t t, wtz_reads_t R, uint32_t id_beg, uint32_t ksize, uint32_t hk, uint32_t ksave, const uint64_t *offs, uint64_t *keys, uint32_t *vals){
	wtz_kfill_f f; f.keys = keys; f.vals = vals; f.pos = offs[t]; f.rid = id_beg + t;
	wtz_kmer_walk(R, id_beg + t, ksize, hk, ksave, f);
}

WTZ_HD uint64_t wtz_run_end(const uint64_t *keys, uint64_t n, uint64_t i){     /* first index > i with a different key */
	uint64_t k = keys[i], step = 1, lo = i, hi;
	while(lo + step < n && keys[lo + step] == k){ lo += step; step <<= 1; }
	hi = WTZ_MIN(lo + step, n);              /* keys[lo]==k, keys[hi]!=k or hi==n */
	while(lo + 1 < hi){ uint64_t mid = lo + (hi - lo) / 2; if(keys[mid] == k) lo = mid; else hi = mid; }
	return hi;
}

/* task over sorted occurrences: at run heads count distinct k-mers (ktyp) and the excess of runs over the
 * reference's saturating 16-bit counter (wtzmo.c:276), so that ktot = n_occ - excess (wtzmo.c:380-388) */
WTZ_HD void wtz_task_kstats(uint64_t i, const uint64_t *keys, uint64_t n, unsigned long long *excess, unsigned long long *ktyp){
	if(i && keys[i - 1] == keys[i]) return;
	uint64_t c = wtz_run_end(keys, n, i) - i;
#if defined(__HIP_DEVICE_COMPILE__)
	if(c > 0xFFFFu) atomicAdd(excess, (unsigned long long)(c - 0xFFFFu));
	atomicAdd(ktyp, 1ull);
#else
	if(c > 0xFFFFu) *excess += c - 0xFFFFu;
	*ktyp += 1;
#endif
}

/* task: run heads with 2 <= cnt <= K are counted (tab == NULL) or inserted into the hash (wtzmo.c:396-411) */
WTZ_HD void wtz_task_kinsert(uint64_t i, const uint64_t *keys, uint64_t n, uint32_t K, wtz_kslot_t *tab, uint64_t cap_mask, unsigned long long *n_kept){
	if(i && keys[i - 1] == keys[i]) return;
	uint64_t c = wtz_run_end(keys, n, i) - i;
	if(c > 0xFFFFu) c = 0xFFFFu;
	if(c > K || c <= 1) return;
	if(tab == NULL){
#if defined(__HIP_DEVICE_COMPILE__)
		atomicAdd(n_kept, 1ull);
#else
		*n_kept += 1;
#endif
		return;
	}
	uint64_t h = wtz_mix64(keys[i]) & cap_mask;
	for(;;){
#if defined(__HIP_DEVICE_COMPILE__)
		unsigned long long old = atomicCAS((unsigned long long*)&tab[h].key, (unsigned long long)WTZ_KEMPTY, (unsigned long long)keys[i]);
#else
		uint64_t old = tab[h].key; if(old == WTZ_KEMPTY) tab[h].key = keys[i];
#endif
		if(old == WTZ_KEMPTY){ tab[h].val = (i << 16) | c; return; }
		h = (h + 1) & cap_mask;
	}
}

WTZ_HD bool wtz_kprobe(const wtz_kslot_t *tab, uint64_t cap_mask, uint64_t mer, uint64_t *off, uint32_t *cnt){
	uint64_t h = wtz_mix64(mer) & cap_mask;
	for(;;){
		wtz_kslot_t s = tab[h];
		if(s.key == mer){ *off = s.val >> 16; *cnt = (uint32_t)(s.val & 0xFFFFu); return true; }
		if(s.key == WTZ_KEMPTY) return false;
		h = (h + 1) & cap_mask;
	}
}

/* ================= K-zidx ================= */
typedef struct {
	/* position order, per read slice [zoff[r], zoff[r+1]) */
	const uint64_t *zoff;
	uint32_t *mer; uint32_t *pos;      /* pos = off<<1 | dir */
	uint16_t *len; uint8_t *ok;        /* ok: occurrence rank of this z-mer inside the read < max_zmer_freq (candidate side cap, hzm_aln.h:208-211) */
	/* (mer, off)-sorted view: index into the slice */
	uint32_t *sidx;
	/* distinct retained z-mers (0 < cnt < max_zmer_freq, hzm_aln.h:107): slice has the same capacity, dn[r] used */
	uint32_t *dmer; uint32_t *dfirst; uint16_t *dcnt; uint32_t *dn;
} wtz_zindex_t;

template<typename F>
WTZ_HD void wtz_zmer_walk(const wtz_reads_t &R, uint32_t rid, uint32_t zsize, uint32_t hz, F &f){
	const uint64_t mask = 0xFFFFFFFFFFFFFFFFULL >> ((32 - zsize) << 1);
	const uint64_t off = R.rdoff[rid]; const uint32_t len = R.rdlen[rid];
	uint32_t ring[16];
	uint64_t kmer = 0, word = 0; uint32_t i = 0, b = 4;
	for(uint32_t j = 0; j < len; j++){
		uint64_t p = off + j;
		if(j == 0 || (p & 31u) == 0) word = R.bits[p >> 5];
		uint32_t c = (uint32_t)((word >> (((~p) & 31u) << 1)) & 3u);
		if(hz && c == b) continue;
		b = c; i++;
		ring[(i - 1) & 15u] = j;
		kmer = ((kmer << 2) | b) & mask;
		if(i < zsize) continue;
		uint64_t rev = wtz_revcomp_kmer(kmer, zsize);
		if(rev == kmer) continue;
		uint32_t dir = rev > kmer ? 0u : 1u;
		uint32_t mer = (uint32_t)(rev > kmer ? kmer : rev);
		uint32_t zo = ring[(i - zsize) & 15u];
		uint32_t zl = (j + 1 - zo > 0xFFFFu) ? 0xFFFFu : j + 1 - zo;
		f(mer, dir, zo, zl);
	}
}

struct wtz_zcount_f { uint32_t n; WTZ_HDM void operator()(uint32_t, uint32_t, uint32_t, uint32_t){ n++; } };
struct wtz_zfill_f { uint32_t *mer, *pos; uint16_t *len; uint64_t *key; uint32_t k;
	WTZ_HDM void operator()(uint32_t m, uint32_t d, uint32_t o, uint32_t l){ mer[k] = m; pos[k] = (o << 1) | d; len[k] = (uint16_t)l; key[k] = ((uint64_t)m << 32) | k; k++; } };

WTZ_HD void wtz_task_zcount(uint32_t r, wtz_reads_t R, uint32_t zsize, uint32_t hz, uint64_t *cnt){
	wtz_zcount_f f; f.n = 0; wtz_zmer_walk(R, r, zsize, hz, f); cnt[r] = f.n;
}

WTZ_HD void wtz_heapsort_u64(uint64_t *a, uint32_t n){
	if(n < 2) return;
	for(uint32_t start = n / 2; start-- > 0; ){
		uint32_t root = start; uint64_t v = a[root];
		for(;;){ uint32_t ch = 2 * root + 1; if(ch >= n) break; if(ch + 1 < n && a[ch + 1] > a[ch]) ch++; if(a[ch] <= v) break; a[root] = a[ch]; root = ch; }
		a[root] = v;
	}
	for(uint32_t end = n - 1; end > 0; end--){
		uint64_t v = a[end]; a[end] = a[0];
		uint32_t root = 0;
		for(;;){ uint32_t ch = 2 * root + 1; if(ch >= end) break; if(ch + 1 < end && a[ch + 1] > a[ch]) ch++; if(a[ch] <= v) break; a[root] = a[ch]; root = ch; }
		a[root] = v;
	}
}

/* task: build all z-mer views of read r. tmpkey is a u64 scratch array parallel to the slices. */
WTZ_HD void wtz_task_zbuild(uint32_t r, wtz_reads_t R, uint32_t zsize, uint32_t hz, uint32_t max_kcnt, wtz_zindex_t Z, uint64_t *tmpkey){
	const uint64_t o = Z.zoff[r]; const uint32_t n = (uint32_t)(Z.zoff[r + 1] - o);
	wtz_zfill_f f; f.mer = Z.mer + o; f.pos = Z.pos + o; f.len = Z.len + o; f.key = tmpkey + o; f.k = 0;
	wtz_zmer_walk(R, r, zsize, hz, f);
	wtz_heapsort_u64(tmpkey + o, n);          /* (mer, position) is a total order: any sort is exact (hzm_aln.h:101) */
	uint32_t nd = 0;
	for(uint32_t i = 0; i < n; ){
		uint32_t m = (uint32_t)(tmpkey[o + i] >> 32), j = i;
		while(j < n && (uint32_t)(tmpkey[o + j] >> 32) == m){
			uint32_t k = (uint32_t)(tmpkey[o + j] & 0xFFFFFFFFu);
			Z.sidx[o + j] = k;
			uint32_t rank = j - i;
			/* u8 counter in the reference: with max_kcnt > 255 the cap can never trigger before the counter wraps */
			Z.ok[o + k] = (max_kcnt > 255u) ? 1 : (rank < max_kcnt ? 1 : 0);
			j++;
		}
		uint32_t c = j - i;
		if(c && c < max_kcnt){ Z.dmer[o + nd] = m; Z.dfirst[o + nd] = i; Z.dcnt[o + nd] = (uint16_t)WTZ_MIN(c, 0xFFFFu); nd++; }
		i = j;
	}
	Z.dn[r] = nd;
}

/* ================= K-seed ================= */
typedef struct { uint32_t key, ol, lst; } wtz_gacc_t;        /* per (rd<<1|dir): running union length */

struct wtz_scount_f { const wtz_kslot_t *tab; uint64_t mask; uint64_t tot; uint64_t nprobe;
	WTZ_HDM void operator()(uint64_t mer, uint32_t, uint32_t, uint32_t){ uint64_t o; uint32_t c; nprobe++; if(wtz_kprobe(tab, mask, mer, &o, &c)) tot += c; } };

struct wtz_sacc_f {
	const wtz_kslot_t *tab; uint64_t mask; const uint32_t *seeds; const uint32_t *rdlen;
	uint32_t pbid, pblen_up; wtz_gacc_t *map; uint32_t mmask; uint32_t nkeys;
	WTZ_HDM void operator()(uint64_t mer, uint32_t, uint32_t qoff, uint32_t qend){
		uint64_t o; uint32_t c;
		if(!wtz_kprobe(tab, mask, mer, &o, &c)) return;
		uint32_t len = qend - qoff; if(len > 0xFFFFu) len = 0xFFFFu;
		for(uint32_t k = 0; k < c; k++){
			uint32_t s = seeds[o + k];
			if((s >> 1) == pbid) continue;                       /* wtzmo.c:488 */
			if(rdlen[s >> 1] > pblen_up) continue;              /* wtzmo.c:489 */
			uint32_t h = (uint32_t)wtz_mix64(s) & mmask;
			while(map[h].key != 0xFFFFFFFFu && map[h].key != s) h = (h + 1) & mmask;
			if(map[h].key == 0xFFFFFFFFu){ map[h].key = s; map[h].ol = 0; map[h].lst = 0; nkeys++; }
			if(qoff >= map[h].lst) map[h].ol += len; else map[h].ol += qoff + len - map[h].lst;   /* wtzmo.c:558-559 */
			map[h].lst = qoff + len;
		}
	}
};

struct wtz_candcmp_f { WTZ_HDM int operator()(uint64_t a, uint64_t b) const { uint32_t x = (uint32_t)a, y = (uint32_t)b; return x > y ? 1 : (x < y ? -1 : 0); } };

#define WTZ_CAND_NONE 0xFFFFFFFF00000000ULL

/* The order-sensitive tail of A3 (wtzmo.c:516-571): strand merge x1/x2 + top-ncand min-heap with its two quirks.
 * groups: (key<<32 | ol) ascending by key. heap: in/out array with room for ncand+1 entries. */
WTZ_HD void wtz_cand_tail(const uint64_t *groups, uint32_t ng, uint32_t kovl, uint32_t ncand, uint64_t *heap, uint32_t *hn){
	uint64_t x1 = WTZ_CAND_NONE, x2; uint32_t n = *hn; wtz_candcmp_f cmp;
	for(uint32_t i = 0; i < ng; i++){
		uint32_t ol = (uint32_t)groups[i], key = (uint32_t)(groups[i] >> 32);
		if(ol < kovl) continue;
		x2 = (((uint64_t)(key >> 1)) << 32) | ol;
		if((x1 >> 32) == (x2 >> 32)){ x1 = (uint32_t)x1 > (uint32_t)x2 ? x1 : x2; }
		else if(x1 == WTZ_CAND_NONE){ x1 = x2; }
		else {
			if(n >= ncand){ if((uint32_t)heap[0] < ol){ heap[0] = x1; wtz_heap_sift(heap, n, 0u, cmp); } }
			else wtz_heap_push(heap, n, x1, cmp);
			x1 = x2;
		}
	}
	if(n >= ncand){ /* compared against ol == 0: never replaces (wtzmo.c:563-567) */ }
	else wtz_heap_push(heap, n, x1, cmp);
	*hn = n;
}

/* task: candidates of query qids[t]; cand_out row stride = ncand + 1 */
WTZ_HD void wtz_task_candidates(uint32_t t, wtz_reads_t R, const uint32_t *qids, const wtz_params_t *P,
		const wtz_kslot_t *tab, uint64_t tmask, const uint32_t *seeds, wtz_pool_t *pool, uint64_t *cand_out, uint32_t *ncand_out, uint32_t stride,
		unsigned long long *algo_bytes){
	const uint32_t pbid = qids[t];
	wtz_scount_f cf; cf.tab = tab; cf.mask = tmask; cf.tot = 0; cf.nprobe = 0;
	wtz_kmer_walk(R, pbid, P->ksize, P->hk, P->ksave, cf);
	{   /* algorithmic bytes of seed lookup: L/4 read + 16 B per probe + 4 B per seed entry */
		unsigned long long b = (unsigned long long)R.rdlen[pbid] / 4 + 16ull * cf.nprobe + 4ull * cf.tot;
#if defined(__HIP_DEVICE_COMPILE__)
		atomicAdd(algo_bytes, b);
#else
		*algo_bytes += b;
#endif
	}
	uint32_t cap = 16; while((uint64_t)cap < cf.tot * 2 + 2) cap <<= 1;
	uint64_t *heap = cand_out + (size_t)t * stride;
	wtz_gacc_t *map = (wtz_gacc_t*)wtz_pool_alloc(pool, (size_t)cap * sizeof(wtz_gacc_t));
	if(map == NULL){ ncand_out[t] = 0xFFFFFFFFu; return; }
	for(uint32_t i = 0; i < cap; i++) map[i].key = 0xFFFFFFFFu;
	wtz_sacc_f af; af.tab = tab; af.mask = tmask; af.seeds = seeds; af.rdlen = R.rdlen; af.pbid = pbid;
	af.pblen_up = (uint32_t)(R.rdlen[pbid] * 1.2);                       /* double multiply, wtzmo.c:445 */
	af.map = map; af.mmask = cap - 1; af.nkeys = 0;
	wtz_kmer_walk(R, pbid, P->ksize, P->hk, P->ksave, af);
	/* compact (key, ol) in place over the map storage, sort by key */
	uint64_t *g = (uint64_t*)wtz_pool_alloc(pool, (size_t)(af.nkeys + 1) * 8);
	if(g == NULL){ ncand_out[t] = 0xFFFFFFFFu; return; }
	uint32_t ng = 0;
	for(uint32_t i = 0; i < cap; i++) if(map[i].key != 0xFFFFFFFFu) g[ng++] = ((uint64_t)map[i].key << 32) | map[i].ol;
	wtz_heapsort_u64(g, ng);
	uint32_t hn = ncand_out[t];                 /* heap carried across index parts (-G), 0 otherwise */
	wtz_cand_tail(g, ng, P->kovl, P->ncand, heap, &hn);
	ncand_out[t] = hn;
}

#endif
