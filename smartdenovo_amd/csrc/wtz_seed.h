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
/* A walk can be cut into independent pieces: the k-mer reported at position j (the first base of its last homopolymer run)
 * depends on the previous `nruns` runs only, so a piece that reports positions [jb, je) starts `nruns` run starts before jb
 * with a cold state and simply does not report until it reaches jb. */
#define WTZ_WALK_CHUNK 1024u
WTZ_HD uint32_t wtz_walk_warm_start(const wtz_reads_t &R, uint32_t rid, uint32_t hp, uint32_t nruns, uint32_t jb){
	if(jb == 0) return 0;
	const uint64_t off = R.rdoff[rid];
	uint32_t j = jb, cnt = 0;
	while(j > 0){
		j--;
		const bool run_start = (j == 0) || !hp || wtz_base_at(R.bits, off + j) != wtz_base_at(R.bits, off + j - 1);
		if(run_start && ++cnt >= nruns) break;
	}
	return j;
}

template<typename F>
WTZ_HD void wtz_kmer_walk(const wtz_reads_t &R, uint32_t rid, uint32_t ksize, uint32_t hk, uint32_t ksave, F &f, uint32_t jb = 0, uint32_t je = 0xFFFFFFFFu){
	const uint64_t mask = 0xFFFFFFFFFFFFFFFFULL >> ((32 - ksize) << 1);
	const uint64_t off = R.rdoff[rid]; const uint32_t len = R.rdlen[rid] < je ? R.rdlen[rid] : je;
	uint32_t ring[32];                      /* start positions of the last ksize hp-runs (hzoff, wtzmo.c:461) */
	uint64_t kmer = 0; uint32_t i = 0; uint32_t b = 4;
	uint64_t word = 0;
	const uint32_t j_first = wtz_walk_warm_start(R, rid, hk, ksize, jb);
	for(uint32_t j = j_first; j < len; j++){
		uint64_t p = off + j;
		if(j == j_first || (p & 31u) == 0) word = R.bits[p >> 5];
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
		if(j < jb) continue;
		f(mer, dir, ring[(i - ksize) & 31u], j + 1);
	}
}

/* ================= K-idx ================= */
struct wtz_kcount_f { uint32_t n; WTZ_HDM void operator()(uint64_t, uint32_t, uint32_t, uint32_t){ n++; } };
struct wtz_kfill_f  { uint64_t *keys; uint32_t *vals; uint64_t pos; uint32_t rid;
	WTZ_HDM void operator()(uint64_t mer, uint32_t dir, uint32_t, uint32_t){ keys[pos] = mer; vals[pos] = (rid << 1) | dir; pos++; } };

/* task: count sampled k-mers of piece t = (read piece_rid[t], positions [piece_jb[t], +WTZ_WALK_CHUNK)) */
WTZ_HD void wtz_task_kcount(uint32_t t, wtz_reads_t R, const uint32_t *piece_rid, const uint32_t *piece_jb, uint32_t ksize, uint32_t hk, uint32_t ksave, uint64_t *cnt){
	wtz_kcount_f f; f.n = 0;
	wtz_kmer_walk(R, piece_rid[t], ksize, hk, ksave, f, piece_jb[t], piece_jb[t] + WTZ_WALK_CHUNK);
	cnt[t] = f.n;
}
/* task: write (k-mer, rd<<1|dir) of piece t at offs[t] (pieces are listed in read order, so the output is the sequential walk's) */
WTZ_HD void wtz_task_kfill(uint32_t t, wtz_reads_t R, const uint32_t *piece_rid, const uint32_t *piece_jb, uint32_t ksize, uint32_t hk, uint32_t ksave, const uint64_t *offs, uint64_t *keys, uint32_t *vals){
	wtz_kfill_f f; f.keys = keys; f.vals = vals; f.pos = offs[t]; f.rid = piece_rid[t];
	wtz_kmer_walk(R, piece_rid[t], ksize, hk, ksave, f, piece_jb[t], piece_jb[t] + WTZ_WALK_CHUNK);
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
	const bool head = !(i && keys[i - 1] == keys[i]);
#if defined(__HIP_DEVICE_COMPILE__)
	{   /* one atomic per wavefront, not per run head: they all hit the same address */
		const unsigned long long m = __ballot(head);
		if(m && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m)) atomicAdd(ktyp, (unsigned long long)__popcll(m));
	}
#else
	if(head) *ktyp += 1;
#endif
	if(!head) return;
	uint64_t c = wtz_run_end(keys, n, i) - i;
#if defined(__HIP_DEVICE_COMPILE__)
	if(c > 0xFFFFu) atomicAdd(excess, (unsigned long long)(c - 0xFFFFu));
#else
	if(c > 0xFFFFu) *excess += c - 0xFFFFu;
#endif
}

/* sharded index: the distinct k-mers of the sorted occurrences with their run lengths, compacted through the scan of the head flags */
WTZ_HD void wtz_task_khead(uint64_t i, const uint64_t *keys, uint32_t *flag){ flag[i] = !(i && keys[i - 1] == keys[i]) ? 1u : 0u; }
WTZ_HD void wtz_task_kdistinct(uint64_t i, const uint64_t *keys, uint64_t n, const uint32_t *flag, const uint32_t *dpos, uint64_t *dk, uint32_t *dc, uint64_t *dstart){
	if(!flag[i]) return;
	const uint64_t c = wtz_run_end(keys, n, i) - i;
	dk[dpos[i]] = keys[i]; dc[dpos[i]] = c > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)c; dstart[dpos[i]] = i;
}
/* sharded index: distinct k-mer d with total count tc over all shards: kept iff 2 <= min(tc, 0xFFFF) <= K (wtzmo.c:276, 401-405) */
WTZ_HD void wtz_task_kinsert_total(uint64_t d, const uint64_t *dk, const uint32_t *dc, const uint64_t *dstart, const uint32_t *tc, uint32_t K, wtz_kslot_t *tab, uint64_t cap_mask, unsigned long long *n_kept){
	uint32_t c = tc[d]; if(c > 0xFFFFu) c = 0xFFFFu;
	const bool kept = !(c > K || c <= 1);
	if(tab == NULL){
#if defined(__HIP_DEVICE_COMPILE__)
		const unsigned long long m = __ballot(kept);
		if(m && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m)) atomicAdd(n_kept, (unsigned long long)__popcll(m));
#else
		if(kept) *n_kept += 1;
#endif
		return;
	}
	if(!kept) return;
	uint64_t h = wtz_mix64(dk[d]) & cap_mask;
	for(;;){
#if defined(__HIP_DEVICE_COMPILE__)
		unsigned long long old = atomicCAS((unsigned long long*)&tab[h].key, (unsigned long long)WTZ_KEMPTY, (unsigned long long)dk[d]);
#else
		uint64_t old = tab[h].key; if(old == WTZ_KEMPTY) tab[h].key = dk[d];
#endif
		if(old == WTZ_KEMPTY){ tab[h].val = (dstart[d] << 16) | (uint64_t)(dc[d] > 0xFFFFu ? 0xFFFFu : dc[d]); return; }
		h = (h + 1) & cap_mask;
	}
}

/* strided forms of the two counting passes: a thread takes every nt-th occurrence and adds its sums once per WAVEFRONT at the end (3.5 M
 * same-address atomics - one per wavefront of the per-occurrence form - serialised in L2: 42 ms each at configs[2] for 2 ms of reading) */
WTZ_HD void wtz_task_kstats_stride(uint64_t t, uint64_t nt, const uint64_t *keys, uint64_t n, unsigned long long *excess, unsigned long long *ktyp){
	unsigned long long heads = 0, exc = 0;
	for(uint64_t i = t; i < n; i += nt){
		if(i && keys[i - 1] == keys[i]) continue;
		heads++;
		const uint64_t c = wtz_run_end(keys, n, i) - i;
		if(c > 0xFFFFu) exc += c - 0xFFFFu;
	}
#if defined(__HIP_DEVICE_COMPILE__)
	for(int d = 32; d > 0; d >>= 1){ heads += __shfl_xor(heads, d, 64); exc += __shfl_xor(exc, d, 64); }
	if((threadIdx.x & 63u) == 0){ if(heads) atomicAdd(ktyp, heads); if(exc) atomicAdd(excess, exc); }
#else
	*ktyp += heads; *excess += exc;
#endif
}
WTZ_HD void wtz_task_kkept_stride(uint64_t t, uint64_t nt, const uint64_t *keys, uint64_t n, uint32_t K, unsigned long long *n_kept){
	unsigned long long kept = 0;
	for(uint64_t i = t; i < n; i += nt){
		if(i && keys[i - 1] == keys[i]) continue;
		uint64_t c = wtz_run_end(keys, n, i) - i; if(c > 0xFFFFu) c = 0xFFFFu;
		if(!(c > K || c <= 1)) kept++;
	}
#if defined(__HIP_DEVICE_COMPILE__)
	for(int d = 32; d > 0; d >>= 1) kept += __shfl_xor(kept, d, 64);
	if((threadIdx.x & 63u) == 0 && kept) atomicAdd(n_kept, kept);
#else
	*n_kept += kept;
#endif
}

/* task: run heads with 2 <= cnt <= K are counted (tab == NULL) or inserted into the hash (wtzmo.c:396-411) */
WTZ_HD void wtz_task_kinsert(uint64_t i, const uint64_t *keys, uint64_t n, uint32_t K, wtz_kslot_t *tab, uint64_t cap_mask, unsigned long long *n_kept){
	const bool head = !(i && keys[i - 1] == keys[i]);
	uint64_t c = 0;
	if(head){ c = wtz_run_end(keys, n, i) - i; if(c > 0xFFFFu) c = 0xFFFFu; }
	const bool kept = head && !(c > K || c <= 1);
	if(tab == NULL){
#if defined(__HIP_DEVICE_COMPILE__)
		const unsigned long long m = __ballot(kept);      /* one atomic per wavefront */
		if(m && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m)) atomicAdd(n_kept, (unsigned long long)__popcll(m));
#else
		if(kept) *n_kept += 1;
#endif
		return;
	}
	if(!kept) return;
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
WTZ_HD void wtz_zmer_walk(const wtz_reads_t &R, uint32_t rid, uint32_t zsize, uint32_t hz, F &f, uint32_t jb = 0, uint32_t je = 0xFFFFFFFFu){
	const uint64_t mask = 0xFFFFFFFFFFFFFFFFULL >> ((32 - zsize) << 1);
	const uint64_t off = R.rdoff[rid]; const uint32_t len = R.rdlen[rid] < je ? R.rdlen[rid] : je;
	uint32_t ring[16];
	uint64_t kmer = 0, word = 0; uint32_t i = 0, b = 4;
	const uint32_t j_first = wtz_walk_warm_start(R, rid, hz, zsize, jb);
	for(uint32_t j = j_first; j < len; j++){
		uint64_t p = off + j;
		if(j == j_first || (p & 31u) == 0) word = R.bits[p >> 5];
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
		if(j < jb) continue;
		f(mer, dir, zo, zl);
	}
}

struct wtz_zcount_f { uint32_t n; WTZ_HDM void operator()(uint32_t, uint32_t, uint32_t, uint32_t){ n++; } };
struct wtz_zfill_f { uint32_t *mer, *pos; uint16_t *len; uint64_t *key; uint32_t k;
	WTZ_HDM void operator()(uint32_t m, uint32_t d, uint32_t o, uint32_t l){ mer[k] = m; pos[k] = (o << 1) | d; len[k] = (uint16_t)l; key[k] = ((uint64_t)m << 32) | k; k++; } };

WTZ_HD void wtz_task_zcount(uint32_t t, wtz_reads_t R, const uint32_t *piece_rid, const uint32_t *piece_jb, uint32_t zsize, uint32_t hz, uint64_t *cnt){
	wtz_zcount_f f; f.n = 0; wtz_zmer_walk(R, piece_rid[t], zsize, hz, f, piece_jb[t], piece_jb[t] + WTZ_WALK_CHUNK); cnt[t] = f.n;
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

/*
 * The z-index of ALL reads is built with device-wide primitives instead of one sort per read:
 *   K_zfill    lane per read   : position-ordered mer / pos / len and the sort key (read<<32 | mer), value = position
 *   radix sort (rocPRIM, stable): by (read, mer); positions stay ascending inside a run  ==  (mer, off) order (hzm_aln.h:101)
 *   K_zrun     lane per element: run heads, run lengths, the candidate-side cap `ok` (occurrence rank < max_kcnt)
 *   exclusive scan of the retained-head flags -> dense index of every distinct z-mer inside its read (the rank_bitvec of
 *   hzm_aln.h:107-114)
 *   K_zdistinct lane per element: dmer / dfirst / dcnt, and dn per read
 */
struct wtz_zfill2_f { uint32_t *mer, *pos; uint16_t *len; uint64_t *key; uint32_t *val; uint32_t k; uint64_t rid;
	WTZ_HDM void operator()(uint32_t m, uint32_t d, uint32_t o, uint32_t l){ mer[k] = m; pos[k] = (o << 1) | d; len[k] = (uint16_t)l; key[k] = (rid << 32) | m; val[k] = k; k++; } };

/* piece t writes at poff[t] (absolute); `val` = position of the z-mer inside its read's list */
/* `base`: the build runs over chunks of consecutive reads (bounded temporaries); key / flag / cnt / dpos are the chunk's arrays and start at element `base` of the index, val (= Z.sidx) and the index arrays are whole */
WTZ_HD void wtz_task_zfill(uint32_t t, wtz_reads_t R, const uint32_t *piece_rid, const uint32_t *piece_jb, const uint64_t *poff, uint32_t zsize, uint32_t hz, wtz_zindex_t Z, uint64_t *key, uint32_t *val, uint64_t base = 0){
	const uint32_t r = piece_rid[t];
	const uint64_t o = Z.zoff[r];
	wtz_zfill2_f f; f.mer = Z.mer + o; f.pos = Z.pos + o; f.len = Z.len + o; f.key = key + (o - base); f.val = val + o; f.k = (uint32_t)(poff[t] - o); f.rid = r;
	wtz_zmer_walk(R, r, zsize, hz, f, piece_jb[t], piece_jb[t] + WTZ_WALK_CHUNK);
}

/* element i of the (read, mer)-sorted array: flag[i] = 1 for the head of a retained run, cnt[i] = its length */
WTZ_HD void wtz_task_zrun(uint64_t i, const uint64_t *key, const uint32_t *val, uint64_t n, uint32_t max_kcnt, wtz_zindex_t Z, uint32_t *flag, uint32_t *cnt){
	const uint64_t k = key[i];
	const uint64_t o = Z.zoff[(uint32_t)(k >> 32)];
	/* occurrence rank inside the run (bounded backward scan): the candidate-side cap of hzm_aln.h:208-211; the reference
	 * counts in a u8, so with max_kcnt > 255 the cap can never trigger before the counter wraps */
	uint32_t rank = 0;
	while(rank < max_kcnt && i > rank && key[i - rank - 1] == k) rank++;
	Z.ok[o + val[i]] = (max_kcnt > 255u) ? 1 : (rank < max_kcnt ? 1 : 0);
	if(i && key[i - 1] == k){ flag[i] = 0; cnt[i] = 0; return; }
	const uint64_t c = wtz_run_end(key, n, i) - i;
	cnt[i] = (uint32_t)WTZ_MIN(c, (uint64_t)0xFFFFu);
	flag[i] = (c && c < max_kcnt) ? 1u : 0u;                 /* kept iff 0 < count < max_kcnt (hzm_aln.h:107) */
}

WTZ_HD void wtz_task_zdistinct(uint64_t i, const uint64_t *key, const uint32_t *flag, const uint32_t *cnt, const uint32_t *dpos, wtz_zindex_t Z, uint64_t base = 0){      /* i: element of the chunk */
	if(!flag[i]) return;
	const uint32_t r = (uint32_t)(key[i] >> 32);
	const uint64_t o = Z.zoff[r];
	const uint32_t d = dpos[i] - dpos[o - base];
	Z.dmer[o + d] = (uint32_t)key[i]; Z.dfirst[o + d] = (uint32_t)(i + base - o); Z.dcnt[o + d] = (uint16_t)cnt[i];
}

WTZ_HD void wtz_task_zdn(uint32_t r, const uint32_t *dpos, wtz_zindex_t Z, uint64_t base = 0){ Z.dn[r] = dpos[Z.zoff[r + 1] - base] - dpos[Z.zoff[r] - base]; }

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
/* the same in two pieces, so that a caller can feed the groups in portions (staged through LDS): `n` entries in the heap, `x1` the candidate not yet pushed */
WTZ_HD void wtz_cand_tail_step(const uint64_t *groups, uint32_t ng, uint32_t kovl, uint32_t ncand, uint64_t *heap, uint32_t &n, uint64_t &x1){
	uint64_t x2; wtz_candcmp_f cmp;
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
}
WTZ_HD void wtz_cand_tail_finish(uint32_t ncand, uint64_t *heap, uint32_t &n, uint64_t x1){
	wtz_candcmp_f cmp;
	if(n >= ncand){ /* compared against ol == 0: never replaces (wtzmo.c:563-567) */ }
	else wtz_heap_push(heap, n, x1, cmp);
}
WTZ_HD void wtz_cand_tail(const uint64_t *groups, uint32_t ng, uint32_t kovl, uint32_t ncand, uint64_t *heap, uint32_t *hn){
	uint64_t x1 = WTZ_CAND_NONE; uint32_t n = *hn;
	wtz_cand_tail_step(groups, ng, kovl, ncand, heap, n, x1);
	wtz_cand_tail_finish(ncand, heap, n, x1);
	*hn = n;
}

/* bitonic sort of np (power of two) u64 words by the whole wavefront; the host emulation uses the in-lane heapsort */
WTZ_HD void wtz_coop_sort_u64(uint64_t *w, uint32_t np){
#if defined(__HIP_DEVICE_COMPILE__)
	const uint32_t lane = WTZ_LANE;
	__threadfence_block();
	for(uint32_t k = 2; k <= np; k <<= 1){
		for(uint32_t j = k >> 1; j > 0; j >>= 1){
			for(uint32_t t0 = 0; t0 < np / 2; t0 += 256){
				uint64_t a[4], b[4]; uint32_t ia[4];
				#pragma unroll
				for(int u = 0; u < 4; u++){
					const uint32_t t = t0 + u * 64 + lane;
					const uint32_t i = ((t / j) * (j << 1)) + (t % j);
					ia[u] = i;
					if(t < np / 2){ a[u] = w[i]; b[u] = w[i + j]; } else { a[u] = 0; b[u] = 0; }
				}
				#pragma unroll
				for(int u = 0; u < 4; u++){
					const uint32_t t = t0 + u * 64 + lane;
					if(t < np / 2){ const bool asc = ((ia[u] & k) == 0); if((a[u] > b[u]) == asc){ w[ia[u]] = b[u]; w[ia[u] + j] = a[u]; } }
				}
			}
			__threadfence_block();
		}
	}
#else
	wtz_heapsort_u64(w, np);
#endif
}

/* Bitonic sort of np (power of two) u64 words that live in HBM, through an LDS window of `ln` words (power of two): chunks
 * of ln words are sorted entirely in LDS, and of every later merge stage only the exchanges at distance >= ln touch HBM; the
 * rest of the stage runs on LDS-resident chunks again.  ~log2(np/ln)^2/2 + 2 log2(np/ln) passes over HBM instead of
 * log2(np)^2/2. */
#if defined(__HIP_DEVICE_COMPILE__)
WTZ_D void wtz_bitonic_lds_steps(uint64_t *a, uint32_t n, uint32_t g0, uint32_t k, uint32_t jstart){
	const uint32_t lane = WTZ_LANE;
	for(uint32_t j = jstart; j > 0; j >>= 1){
		for(uint32_t t0 = 0; t0 < n / 2; t0 += 256){
			uint64_t x[4], y[4]; uint32_t ia[4];
			#pragma unroll
			for(int u = 0; u < 4; u++){
				const uint32_t t = t0 + u * 64 + lane;
				const uint32_t i = ((t / j) * (j << 1)) + (t % j);
				ia[u] = i;
				if(t < n / 2){ x[u] = a[i]; y[u] = a[i + j]; } else { x[u] = 0; y[u] = 0; }
			}
			#pragma unroll
			for(int u = 0; u < 4; u++){
				const uint32_t t = t0 + u * 64 + lane;
				if(t < n / 2){ const bool asc = (((g0 + ia[u]) & k) == 0); if((x[u] > y[u]) == asc){ a[ia[u]] = y[u]; a[ia[u] + j] = x[u]; } }
			}
		}
		__threadfence_block();
	}
}
#endif
WTZ_HD void wtz_coop_sort_u64_windowed(uint64_t *w, uint32_t np, uint64_t *lds, uint32_t ln){
#if defined(__HIP_DEVICE_COMPILE__)
	const uint32_t lane = WTZ_LANE;
	if(lds == NULL || ln < 128u){ wtz_coop_sort_u64(w, np); return; }
	__threadfence_block();
	if(np <= ln){
		for(uint32_t i = lane; i < np; i += 64) lds[i] = w[i];
		__threadfence_block();
		for(uint32_t k = 2; k <= np; k <<= 1) wtz_bitonic_lds_steps(lds, np, 0, k, k >> 1);
		for(uint32_t i = lane; i < np; i += 64) w[i] = lds[i];
		__threadfence_block();
		return;
	}
	for(uint32_t c0 = 0; c0 < np; c0 += ln){        /* chunks sorted in LDS (direction of the last stage from the global index) */
		for(uint32_t i = lane; i < ln; i += 64) lds[i] = w[c0 + i];
		__threadfence_block();
		for(uint32_t k = 2; k <= ln; k <<= 1) wtz_bitonic_lds_steps(lds, ln, c0, k, k >> 1);
		for(uint32_t i = lane; i < ln; i += 64) w[c0 + i] = lds[i];
		__threadfence_block();
	}
	for(uint32_t k = ln << 1; k <= np; k <<= 1){
		for(uint32_t j = k >> 1; j >= ln; j >>= 1){   /* long-distance exchanges in HBM */
			for(uint32_t t = lane; t < np / 2; t += 64){
				const uint32_t i = ((t / j) * (j << 1)) + (t % j);
				const uint64_t x = w[i], y = w[i + j];
				const bool asc = ((i & k) == 0);
				if((x > y) == asc){ w[i] = y; w[i + j] = x; }
			}
			__threadfence_block();
		}
		for(uint32_t c0 = 0; c0 < np; c0 += ln){    /* the rest of the stage inside LDS */
			for(uint32_t i = lane; i < ln; i += 64) lds[i] = w[c0 + i];
			__threadfence_block();
			wtz_bitonic_lds_steps(lds, ln, c0, k, ln >> 1);
			for(uint32_t i = lane; i < ln; i += 64) w[c0 + i] = lds[i];
			__threadfence_block();
		}
	}
#else
	(void)lds; (void)ln;
	wtz_heapsort_u64(w, np);
#endif
}

/* the same network over 32-bit words (LDS-resident band keys of the dot-matrix engine) */
WTZ_HD void wtz_coop_sort_u32(uint32_t *w, uint32_t np){
#if defined(__HIP_DEVICE_COMPILE__)
	const uint32_t lane = WTZ_LANE;
	__threadfence_block();
	for(uint32_t k = 2; k <= np; k <<= 1){
		for(uint32_t j = k >> 1; j > 0; j >>= 1){
			for(uint32_t t0 = 0; t0 < np / 2; t0 += 256){
				uint32_t a[4], b[4], ia[4];
				#pragma unroll
				for(int u = 0; u < 4; u++){
					const uint32_t t = t0 + u * 64 + lane;
					const uint32_t i = ((t / j) * (j << 1)) + (t % j);
					ia[u] = i;
					if(t < np / 2){ a[u] = w[i]; b[u] = w[i + j]; } else { a[u] = 0; b[u] = 0; }
				}
				#pragma unroll
				for(int u = 0; u < 4; u++){
					const uint32_t t = t0 + u * 64 + lane;
					if(t < np / 2){ const bool asc = ((ia[u] & k) == 0); if((a[u] > b[u]) == asc){ w[ia[u]] = b[u]; w[ia[u] + j] = a[u]; } }
				}
			}
			__threadfence_block();
		}
	}
#else
	for(uint32_t i = 1; i < np; i++){ uint32_t v = w[i], q = i; while(q && w[q - 1] > v){ w[q] = w[q - 1]; q--; } w[q] = v; }
#endif
}

#define WTZ_CAND_SKETCH 2048u
#define WTZ_CAND_TAB 2048u
#define WTZ_CAND_OUT 512u
#define WTZ_CAND_STREAM_LDS_BYTES(ncand) ((WTZ_CAND_SKETCH + 3u * WTZ_CAND_TAB) * 4u + WTZ_CAND_OUT * 8u + ((ncand) + 2u) * 8u)
#if defined(__HIP_DEVICE_COMPILE__)
/*
 * Phases C-E of the candidate task without expanding and sorting the tuples.
 *
 * The reference's k-way merge (wtzmo.c:500-560) delivers the tuples of one (read, strand) group in query-offset order and folds them with
 *     ol += (qoff >= lst) ? len : qoff + len - lst;   lst = qoff + len            (u32; a later, shorter k-mer moves lst back)
 * and only groups with ol >= -d (kovl) ever matter (wtzmo.c:523).  Two facts make a sort-free form exact:
 *   (1) by induction ol >= len of the last tuple > 0 and every step adds at most len, so 0 < ol <= SUM(len): a group whose lengths sum to
 *       less than kovl can be dropped without looking at its order.  Most groups of a query are such chance hits of erroneous k-mers
 *       (~1 500 groups per 10 kb query, ~200 of them real) - a table of ALL groups does not fit LDS, a table of the survivors does;
 *   (2) the sampled k-mers of the query are listed in query-offset order and the seed run of one k-mer is sorted by (read, strand): walking
 *       the k-mers in order with the lanes side by side on the entries of a run gives every group its tuples in the reference's order.
 * Pass 1 (order-free, every lane its own k-mers): LDS sketch  S[hash(read,strand)] += len.  Pass 2 (64 k-mers at a time, every lane its own):
 * does the run have an entry with S >= kovl?  Pass 3 (those k-mers only, in order, 8 runs in flight): fold the surviving entries into an
 * LDS hash table of (lst, ol).  Then the groups with ol >= kovl are compacted, put in (read, strand) order and lane 0 replays the strand merge and
 * the candidate heap (staged in LDS).  Returns false (nothing written) when the survivors overflow the table: the caller then runs the
 * sorting form.
 * id_thr: when the indexed reads are in non-increasing length order (always, unless -b clipped them after the sort) "longer than 1.2 x
 * the query" (wtzmo.c:489) is "read id below id_thr" - no length load per run entry; 0xFFFFFFFF = look the length up.
 */
WTZ_D bool wtz_cand_stream(uint32_t t, const wtz_reads_t &R, uint32_t pbid, uint32_t pblen_up, const wtz_params_t *P, const uint32_t *seeds,
		uint32_t nk, const uint64_t *koff, const uint32_t *kqoff, const uint32_t *kqlen, uint64_t *cand_out, uint32_t *ncand_out, uint32_t stride, uint32_t *lds32, uint32_t id_thr){
	const uint32_t lane = WTZ_LANE, EMPTY = 0xFFFFFFFFu, MASK = WTZ_CAND_TAB - 1u, SMASK = WTZ_CAND_SKETCH - 1u;
	uint32_t *sk = lds32, *keys = sk + WTZ_CAND_SKETCH, *lsts = keys + WTZ_CAND_TAB, *ols = lsts + WTZ_CAND_TAB;
	uint64_t *out = (uint64_t*)(ols + WTZ_CAND_TAB), *heap = out + WTZ_CAND_OUT;
	const uint32_t kovl = P->kovl;
	if(P->ncand + 1u > 1024u) return false;
	for(uint32_t i = lane; i < WTZ_CAND_SKETCH; i += 64) sk[i] = 0u;
	for(uint32_t i = lane; i < WTZ_CAND_TAB; i += 64) keys[i] = EMPTY;
	__threadfence_block();
	auto dropped = [&](uint32_t sd) -> bool {                                  /* wtzmo.c:488-489 */
		if((sd >> 1) == pbid) return true;
		if(id_thr != 0xFFFFFFFFu) return (sd >> 1) < id_thr;
		return R.rdlen[sd >> 1] > pblen_up;
	};
	/* ---- pass 1: length sums per (read, strand) hash ---- */
	for(uint32_t e = lane; e < nk; e += 64){
		const uint64_t oc = koff[e]; const uint32_t c = (uint32_t)(oc & 0xFFFFu); const uint64_t o = oc >> 16;
		const uint32_t ql = kqlen[e] < kovl ? kqlen[e] : kovl;                 /* clamped: the sums cannot wrap */
		uint32_t prev = EMPTY;
		for(uint32_t k = 0; k < c; k++){
			const uint32_t sd = seeds[o + k];
			if(sd != prev && !dropped(sd)) atomicAdd(&sk[(uint32_t)wtz_mix64(sd) & SMASK], ql);       /* an identical neighbour adds 0 to ol */
			prev = sd;
		}
	}
	__threadfence_block();
	uint32_t ngrp = 0;
	/* one run entry per lane: group lookup / insert and the fold of this k-mer's interval */
	auto fold = [&](uint32_t sd, uint32_t prev, bool in_run, uint32_t qo, uint32_t ql){
		bool act = in_run && sd != prev && !dropped(sd) && sk[(uint32_t)wtz_mix64(sd) & SMASK] >= kovl;
		bool fresh = false;
		if(act){
			uint32_t h = (uint32_t)wtz_mix64(sd * 0x9E3779B1u) & MASK;
			for(;;){
				const uint32_t old = atomicCAS(&keys[h], EMPTY, sd);
				if(old == EMPTY){ fresh = true; break; }
				if(old == sd) break;
				h = (h + 1u) & MASK;
			}
			uint32_t lst = fresh ? 0u : lsts[h], ol = fresh ? 0u : ols[h];
			if(qo >= lst) ol += ql; else ol += qo + ql - lst;                     /* wtzmo.c:558-559 */
			lsts[h] = qo + ql; ols[h] = ol;
		}
		ngrp += (uint32_t)__popcll(__ballot(fresh));
		__threadfence_block();
	};
	/* ---- pass 3 ---- */
	constexpr int PF = 8;                                /* runs whose first 64 entries are in flight together: the walk is a chain of dependent loads otherwise */
	for(uint32_t e0 = 0; e0 < nk; e0 += 64){
		const uint32_t e = e0 + lane;
		const uint64_t oc = e < nk ? koff[e] : 0ull;
		uint32_t my_c = (uint32_t)(oc & 0xFFFFu); const uint32_t my_olo = (uint32_t)(oc >> 16), my_ohi = (uint32_t)(oc >> 48);
		const uint32_t my_q = e < nk ? kqoff[e] : 0u, my_l = e < nk ? kqlen[e] : 0u;
		{   /* pass 2, every lane its own k-mer: a run without an entry that can reach -d is not walked */
			bool any = false;
			for(uint32_t k = 0; k < my_c && !any; k++){ const uint32_t sd = seeds[(oc >> 16) + k]; if(!dropped(sd) && sk[(uint32_t)wtz_mix64(sd) & SMASK] >= kovl) any = true; }
			if(!any) my_c = 0;
		}
		unsigned long long hits = __ballot(my_c != 0);
		while(hits){
			int ls[PF]; uint32_t cs[PF], sdv[PF]; uint64_t os[PF];
			#pragma unroll
			for(int g = 0; g < PF; g++){
				ls[g] = -1; cs[g] = 0; os[g] = 0; sdv[g] = EMPTY;
				if(hits){
					const int l = __builtin_ctzll(hits); hits &= hits - 1ull; ls[g] = l;
					cs[g] = (uint32_t)__builtin_amdgcn_readlane((int)my_c, l);
					os[g] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)my_ohi, l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)my_olo, l);
					if(lane < cs[g]) sdv[g] = seeds[os[g] + lane];
				}
			}
			#pragma unroll
			for(int g = 0; g < PF; g++){
				if(ls[g] < 0) continue;                      /* uniform */
				const uint32_t c = cs[g], qo = (uint32_t)__builtin_amdgcn_readlane((int)my_q, ls[g]), ql = (uint32_t)__builtin_amdgcn_readlane((int)my_l, ls[g]);
				if(ngrp + 64u > WTZ_CAND_TAB - WTZ_CAND_TAB / 4u){ WTZ_PROF_CNT(46, 1000000); return false; }      /* more survivors than the table takes at load 3/4 */
				uint32_t sd = sdv[g];
				uint32_t prev = (uint32_t)__shfl_up((int)sd, 1, 64); if(lane == 0) prev = EMPTY;
				uint32_t carry = (uint32_t)__builtin_amdgcn_readlane((int)sd, 63);       /* last entry of the previous 64 of this run */
				fold(sd, prev, lane < c, qo, ql);
				for(uint32_t k0 = 64; k0 < c; k0 += 64){
					if(ngrp + 64u > WTZ_CAND_TAB - WTZ_CAND_TAB / 4u){ WTZ_PROF_CNT(46, 1000000); return false; }
					const uint32_t k = k0 + lane;
					sd = k < c ? seeds[os[g] + k] : EMPTY;
					prev = (uint32_t)__shfl_up((int)sd, 1, 64); if(lane == 0) prev = carry;
					carry = (uint32_t)__builtin_amdgcn_readlane((int)sd, 63);
					fold(sd, prev, k < c, qo, ql);
				}
			}
		}
	}
	/* groups that reach -d, in (read, strand) order */
	uint32_t ng = 0;
	for(uint32_t i0 = 0; i0 < WTZ_CAND_TAB; i0 += 64){
		const uint32_t i = i0 + lane;
		const bool keep = keys[i] != EMPTY && ols[i] >= kovl;
		uint32_t tot; const uint32_t pos = wtz_coop_rank(keep, &tot);
		if(ng + tot > WTZ_CAND_OUT){ WTZ_PROF_CNT(47, 1000000); return false; }
		if(keep) out[ng + pos] = ((uint64_t)keys[i] << 32) | ols[i];
		ng += tot;
	}
	uint32_t np = 64; while(np < ng) np <<= 1;
	for(uint32_t i = ng + lane; i < np; i += 64) out[i] = ~0ull;
	__threadfence_block();
	wtz_coop_sort_u64(out, np);
	/* strand merge + candidate heap with its quirks (wtzmo.c:516-571) on lane 0, the heap row staged in LDS */
	uint32_t hn = ncand_out[t];
	uint64_t *row = cand_out + (size_t)t * stride;
	for(uint32_t i = lane; i < hn; i += 64) heap[i] = row[i];
	__threadfence_block();
	if(lane == 0) wtz_cand_tail(out, ng, kovl, P->ncand, heap, &hn);
	hn = wtz_coop_bcast32(hn);
	__threadfence_block();
	for(uint32_t i = lane; i < hn; i += 64) row[i] = heap[i];
	if(lane == 0) ncand_out[t] = hn;
	WTZ_PROF_CNT(10, 1000000); WTZ_PROF_CNT(11, ngrp * 1000ull); WTZ_PROF_CNT(12, ng * 1000ull);
	return true;
}
#endif

/* ================= K-seed, workgroup form: partition by target read, sort each bucket in LDS =================
 *
 * What the numbers of configs[2] say (tools/analysis/seedstats.c, 100x coverage of a 12 Mbp genome): a query of 15-50 kb has 3-10 k sampled
 * k-mers that hit, 100-340 k seed entries (tuples), and 60-160 k DISTINCT (read, strand) groups of which only 100-650 reach -d: the
 * hp-compressed 16-mers of a 12 Mbp genome collide by chance, so nearly every group is a chance hit of one or two k-mers.  One wavefront
 * per query ordered all tuples with a bitonic network through a 16 KB LDS window: ~700 bytes of HBM traffic per 4-byte seed entry (PMC:
 * 830 GB per step against 5 GB algorithmic), 0.1 % of the HBM roof.  No LDS table can hold 100 k groups, and a sketch in front of it
 * saturates at that density (the streaming form above fell back for a third of the long queries).
 *
 * This form moves every tuple through HBM exactly once, as an MSD radix step by target read:
 *   A  the workgroup's threads walk one piece of the read each and list its sampled k-mers (exact restart: wtz_walk_warm_start)
 *   B  one hash probe per k-mer: seed run (start, length)
 *   H  histogram of the tuples over 4096 bins of the target key (read << 1 | strand), in LDS; self hits and reads longer than 1.2 x
 *      the query are dropped here (wtzmo.c:488-489)
 *   S  exclusive scan -> bin offsets; second pass over the runs scatters (key << 32 | k-mer index) into the pool: the bins come out
 *      contiguous and in key order
 *   P  consecutive bins are grouped into buckets of <= WTZ_CWG_CAP tuples; each bucket is sorted in LDS (bitonic, by key then k-mer
 *      index = query-offset order, the order the reference's k-way merge delivers a group in, wtzmo.c:44-57); then the union-length
 *      recurrence of wtzmo.c:558-560 - ol += (qoff >= lst) ? len : qoff + len - lst; lst = qoff + len - is evaluated WITHOUT a serial
 *      walk: after the first tuple lst is simply the end of the previous tuple, so every tuple's addend depends on its predecessor only
 *      and ol is a segmented sum (u32 arithmetic like the reference's, the wrap-around of a "negative" addend included)
 *   F  thread 0 replays strand merge + candidate heap with its quirks over the groups that reach -d (wtz_cand_tail), unchanged.
 * HBM traffic per tuple: 4 B seed entry read twice, 8 B written and read once: ~24 B instead of ~700.
 */
#ifndef WTZ_CWG_THREADS
#define WTZ_CWG_THREADS 512u
#endif
#define WTZ_CWG_BINS 4096u
#define WTZ_CWG_CAP 2048u
#define WTZ_CWG_LDS_BYTES (64u * 4u + 65536u)      /* scratch words + the larger of { bins + sort slots + interval arrays = 48 KB, the group sketch = 64 KB } */
#if defined(__HIP_DEVICE_COMPILE__)
#define WTZ_WG_TID ((uint32_t)threadIdx.x)
#define WTZ_WG_N ((uint32_t)blockDim.x)
#define WTZ_WG_SYNC() __syncthreads()
#define WTZ_LDS_ADD32(p, v) atomicAdd((p), (v))
#define WTZ_LDS_CAS32(p, o, n) atomicCAS((p), (o), (n))
#else
#define WTZ_WG_TID 0u
#define WTZ_WG_N 1u
#define WTZ_WG_SYNC() do {} while(0)
static inline uint32_t wtz_lds_add32_host(uint32_t *p, uint32_t v){ const uint32_t o = *p; *p += v; return o; }
#define WTZ_LDS_ADD32(p, v) wtz_lds_add32_host((p), (v))
static inline uint32_t wtz_lds_cas32_host(uint32_t *p, uint32_t o, uint32_t n){ const uint32_t c = *p; if(c == o) *p = n; return c; }
#define WTZ_LDS_CAS32(p, o, n) wtz_lds_cas32_host((p), (o), (n))
#endif
/* exclusive prefix sum of one value per thread over the workgroup (tmp: 64 LDS words), *total = the sum */
WTZ_HD uint32_t wtz_wg_excl_scan(uint32_t v, uint32_t *tmp, uint32_t *total){
#if defined(__HIP_DEVICE_COMPILE__)
	uint32_t wtot; const uint32_t ex = wtz_coop_excl_scan(v, &wtot);
	const uint32_t wv = threadIdx.x >> 6, nw = (blockDim.x + 63u) >> 6;
	__syncthreads();
	if((threadIdx.x & 63u) == 0) tmp[wv] = wtot;
	__syncthreads();
	uint32_t base = 0, tot = 0;
	for(uint32_t k = 0; k < nw; k++){ const uint32_t x = tmp[k]; if(k < wv) base += x; tot += x; }
	*total = tot;
	return base + ex;
#else
	(void)tmp; *total = v; return 0;
#endif
}
/* bitonic sort of np (power of two) u64 words in LDS by the whole workgroup */
WTZ_HD void wtz_wg_sort_u64(uint64_t *a, uint32_t np){
#if defined(__HIP_DEVICE_COMPILE__)
	const uint32_t tid = threadIdx.x, nt = blockDim.x;
	for(uint32_t lk = 1; (1u << lk) <= np; lk++){
		for(uint32_t lj = lk; lj-- > 0; ){
			const uint32_t j = 1u << lj;
			__syncthreads();
			for(uint32_t t = tid; t < np / 2; t += nt){
				const uint32_t i = ((t >> lj) << (lj + 1)) | (t & (j - 1u));      /* shifts: j is a power of two, but only the loop knows */
				const uint64_t x = a[i], y = a[i + j];
				const bool asc = ((i >> lk) & 1u) == 0;
				if((x > y) == asc){ a[i] = y; a[i + j] = x; }
			}
		}
	}
	__syncthreads();
#else
	wtz_heapsort_u64(a, np);
#endif
}

typedef struct { uint32_t qoff, qlen; } wtz_kq_t;
/* Group sketch (round 6): 65 536 saturating 8-BIT counters of the groups' length sums in units of 2 bases, 64 KB of LDS (everything behind the scratch words: bins, sort
 * slots and interval arrays are idle while the tuples are walked; the bin histogram is made from the LISTED tuples afterwards).  History: round 5 had 16 384 16-bit
 * counters over 32 KB - at the configs[3] shape (a query of 10 kb meets ~460 000 tuples there: hp-compressed 16-mers have only 4 * 3^15 values, 10 Gbp of reads put
 * dozens of chance occurrences behind every one) every counter reached the threshold and everything passed; the first round-6 form (65 536 4-bit counters in units
 * of 20 bases, 32 KB) still let 51 % through (phase clock at that shape, profiles/r06_seed_lookup_phase_clock_fly70.txt: the round-up of a 21-base tuple to two units
 * put 12 units of a threshold of 15 on the average counter), and sort + fold of what passed were 82 % of the kernel.  Here the average counter holds 37 units of a
 * threshold of 150. */
#define WTZ_CWG_SKETCH 65536u
/* second level: from how many listed tuples on, and how many units one part of it may hold (the host emulation sets both small so that the CPU suite's little inputs go
 * through the second level and through several parts of it) */
#ifdef WTZ_EMUL
#define WTZ_CWG_L2_MIN 48u
#define WTZ_CWG_L2_PART_UNITS 1024u
#else
#define WTZ_CWG_L2_MIN (2u * WTZ_CWG_CAP)
#define WTZ_CWG_L2_PART_UNITS (65536u * 48u)
#endif
WTZ_HD uint32_t wtz_cwg_sk_hash(uint32_t sd){ return (sd * 0x9E3779B1u) >> 16; }      /* 16 bits */
WTZ_HD uint32_t wtz_cwg_sk_hash2(uint32_t sd){ return ((sd ^ (sd >> 15)) * 0x85EBCA6Bu) >> 16; }      /* the second level's: independent of the first */
WTZ_HD uint32_t wtz_cwg_sk_hash3(uint32_t sd){ return ((sd ^ (sd >> 13)) * 0x27D4EB2Fu) >> 16; }      /* which part of the second level a key belongs to */
/* counter h of sk += v, saturating at 255; nothing is added once it holds `thr` (<= 255).  A compare-and-swap loop on the word: racing adds can neither carry into the
 * neighbouring byte nor wrap this one (a wrapped counter would drop a group that reaches -d: a wrong result, not a slow one) */
WTZ_HD void wtz_cwg_sk_add(uint32_t *sk, uint32_t h, uint32_t v, uint32_t thr){
	uint32_t *w = &sk[h >> 2]; const uint32_t sh = (h & 3u) << 3;
	uint32_t cur = *w;
	for(;;){
		const uint32_t c = (cur >> sh) & 255u;
		if(c >= thr) return;
		uint32_t nc = c + v; if(nc > 255u) nc = 255u;
		const uint32_t want = (cur & ~(255u << sh)) | (nc << sh);
		const uint32_t got = WTZ_LDS_CAS32(w, cur, want);
		if(got == cur) return;
		cur = got;
	}
}
WTZ_HD bool wtz_cwg_sk_pass(const uint32_t *sk, uint32_t h, uint32_t thr){ return ((sk[h >> 2] >> ((h & 3u) << 3)) & 255u) >= thr; }
/* inclusive running maximum of one value per thread over the workgroup (tmp: 64 LDS words), *total = the maximum */
WTZ_HD uint32_t wtz_wg_incl_max(uint32_t v, uint32_t *tmp, uint32_t *total){
#if defined(__HIP_DEVICE_COMPILE__)
	const uint32_t in = wtz_coop_incl_max32(v);
	const uint32_t wv = threadIdx.x >> 6, nw = (blockDim.x + 63u) >> 6;
	__syncthreads();
	if((threadIdx.x & 63u) == 63u) tmp[wv] = in;
	__syncthreads();
	uint32_t base = 0, tot = 0;
	for(uint32_t k = 0; k < nw; k++){ const uint32_t x = tmp[k]; if(k < wv && x > base) base = x; if(x > tot) tot = x; }
	*total = tot;
	return in > base ? in : base;
#else
	(void)tmp; *total = v; return v;
#endif
}
/*
 * The z-mer index of ONE read by one workgroup (round 5; hzm_aln.h:70-115 is per read too).  The device-wide form above moves every z-mer of the read set
 * through HBM five times - K_zfill's strided stores (a lane per 1 024-base piece: 64 lanes x 5 arrays of open lines per wave, 480 GB/s), a 49-bit radix sort
 * of (read, mer) keys (7 passes), K_zrun, a scan, K_zdistinct: 131 of the 137 ms the index of configs[2] takes.  A read has a few thousand z-mers: they fit
 * the LDS of one CU.  Here the workgroup walks the read in pieces of 128 bases (two walks: count, then fill - the position-ordered arrays go out as the walk
 * produces them), keeps (mer << 32 | position rank) in LDS, orders them with the workgroup's bitonic network (keys are unique, so the order is the stable
 * (mer, position) order of the radix sort), and derives the sorted view, the candidate-side cap and the table of distinct retained z-mers from the
 * ordered keys with two running scans: nothing is written that is not part of the index.  Reads with more z-mers than the LDS holds stay with the form above.
 *   lds: np u64 keys | bucket cursors | piece offsets | 64 words of scan scratch | the read's 2-bit words (wtz_zr_lds_bytes)
 */
#define WTZ_ZR_SUB 32u                 /* bases per walk piece: a 10 kb read keeps 320 of the workgroup's threads busy (128-base pieces: 80; the walks are what a read's time is made of) */
#define WTZ_ZR_MAXN 16384u
#define WTZ_ZR_SMALL 24u              /* buckets up to this size are ordered by one lane (insertion) */
#define WTZ_ZR_MAXLEN(np) ((np) * 2u)                     /* bases of a read of the class: its packed bases are staged in LDS (the walks' warm starts step BACKWARDS base by base) */
#define WTZ_ZR_MAXPC(np) (WTZ_ZR_MAXLEN(np) / WTZ_ZR_SUB)
WTZ_HD uint32_t wtz_zr_nbk(uint32_t np){ uint32_t b = 64; while(b < np / 4u) b <<= 1; return b; }      /* buckets: a power of two, ~4 z-mers each */
WTZ_HD uint32_t wtz_zr_lds_bytes(uint32_t np){ return np * 8u + (wtz_zr_nbk(np) + 1u) * 4u + WTZ_ZR_MAXPC(np) * 4u + 64u * 4u + (WTZ_ZR_MAXLEN(np) / 32u + 4u) * 8u + 16u * 4u + 64u; }
/* n (any number of) u64 words in LDS ordered by the whole workgroup: the bitonic network with all comparators pointing up (first step of a merge against the
 * mirror image), so that the missing elements behind n act as +infinity without being stored */
WTZ_HD void wtz_wg_sort_u64_n(uint64_t *a, uint32_t n){
#if defined(__HIP_DEVICE_COMPILE__)
	const uint32_t tid = threadIdx.x, nt = blockDim.x;
	uint32_t lp = 1; while((1u << lp) < n) lp++;
	for(uint32_t lk = 1; lk <= lp; lk++){
		for(uint32_t lj = lk; lj-- > 0; ){
			const uint32_t j = 1u << lj;
			__syncthreads();
			for(uint32_t t = tid; t < (1u << lp) / 2; t += nt){
				const uint32_t i = ((t >> lj) << (lj + 1)) | (t & (j - 1u));
				const uint32_t q = (lj + 1 == lk) ? (i ^ ((2u << lj) - 1u)) : (i | j);      /* mirror inside the block of 2j for the first step of a merge, i + j after it */
				if(q < n){ const uint64_t x = a[i], y = a[q]; if(x > y){ a[i] = y; a[q] = x; } }
			}
		}
	}
	__syncthreads();
#else
	wtz_heapsort_u64(a, n);
#endif
}
WTZ_HD void wtz_task_zread(uint32_t r, const wtz_reads_t &R, uint32_t zsize, uint32_t hz, uint32_t zcut, wtz_zindex_t Z, uint32_t *lds, uint32_t np){
	const uint32_t tid = WTZ_WG_TID, nt = WTZ_WG_N;
	const uint64_t o = Z.zoff[r]; const uint32_t n = (uint32_t)(Z.zoff[r + 1] - o);
	if(n == 0){ if(tid == 0) Z.dn[r] = 0; return; }
	/* Round 6: no buckets.  Round 5 dropped the keys into LDS buckets by "the leading bits of the z-mer" - with a shift taken from a 32-bit key, so that every z-mer
	 * of a read (2 * zsize = 20 bits at -z 10) went to bucket 0: 2 n same-address LDS atomics and the whole array through the workgroup's network.
	 * With the buckets really populated (a 20-bit shift) the index of configs[2] took 93 ms instead of 83: hp-compressed canonical z-mers crowd a few leading-bit
	 * prefixes, and every bucket beyond 24 keys is a workgroup sort of its own.  What was fast about the accident is kept and the atomics go: a key's slot is its
	 * position rank, which the fill walk has anyway (no histogram, no cursors), and the whole array is ordered by the network. */
	const uint32_t nbk = wtz_zr_nbk(np), maxpc = WTZ_ZR_MAXPC(np);      /* nbk: the (now unused) cursor words keep the LDS layout of wtz_zr_lds_bytes */
	uint64_t *keys = (uint64_t*)lds; uint32_t *bk = lds + 2 * (size_t)np, *pc = bk + nbk + 1, *tmp = pc + maxpc;
	const uint32_t len = R.rdlen[r], npc = (len + WTZ_ZR_SUB - 1) / WTZ_ZR_SUB;
	/* the read's 2-bit words in LDS, addressed as a one-read bank: read 0 starts at the offset of the read inside its first word */
	uint64_t *lbits = (uint64_t*)(((uintptr_t)(tmp + 64) + 7u) & ~(uintptr_t)7u);
	const uint32_t nwl = WTZ_ZR_MAXLEN(np) / 32u + 4u;
	uint64_t *loff = lbits + nwl; uint32_t *llen = (uint32_t*)(loff + 1);
	{
		const uint64_t off = R.rdoff[r], w0 = off >> 5, nw = ((off + len + 31u) >> 5) - w0;
		for(uint32_t w = tid; w < (uint32_t)nw; w += nt) lbits[w] = R.bits[w0 + w];
		if(tid == 0){ loff[0] = off & 31u; llen[0] = len; }
	}
	wtz_reads_t RL; RL.bits = lbits; RL.rdoff = loff; RL.rdlen = llen; RL.n_reads = 1;
	WTZ_WG_SYNC();
	for(uint32_t p = tid; p < npc; p += nt){ wtz_zcount_f f; f.n = 0; wtz_zmer_walk(RL, 0u, zsize, hz, f, p * WTZ_ZR_SUB, p * WTZ_ZR_SUB + WTZ_ZR_SUB); pc[p] = f.n; }
	WTZ_WG_SYNC();
	{
		uint32_t carry = 0;
		for(uint32_t b = 0; b < npc; b += nt){
			const uint32_t v = b + tid < npc ? pc[b + tid] : 0u;
			uint32_t tot; const uint32_t ex = wtz_wg_excl_scan(v, tmp, &tot);
			if(b + tid < npc) pc[b + tid] = carry + ex;
			carry += tot;
			WTZ_WG_SYNC();
		}
	}
	for(uint32_t p = tid; p < npc; p += nt){
		wtz_zfill_f f; f.mer = Z.mer + o; f.pos = Z.pos + o; f.len = Z.len + o; f.key = keys; f.k = pc[p];       /* key[k] = mer << 32 | k: unique, so every correct order IS the stable (mer, position) order */
		wtz_zmer_walk(RL, 0u, zsize, hz, f, p * WTZ_ZR_SUB, p * WTZ_ZR_SUB + WTZ_ZR_SUB);
	}
	WTZ_WG_SYNC();
	wtz_wg_sort_u64_n(keys, n);
	WTZ_WG_SYNC();
	/* ordered keys -> sorted view, cap flags, distinct table.  Element i: head = first of its run of equal z-mers, tail = last; the run's length is known at
	 * its tail (i - head + 1), and numbering the retained runs by their tails gives the same dense index as numbering them by their heads */
	uint32_t hcarry = 0, dcarry = 0;
	for(uint32_t b = 0; b < n; b += nt){
		const uint32_t i = b + tid; const bool in = i < n;
		const uint64_t k = in ? keys[i] : 0ull;
		const uint32_t m = (uint32_t)(k >> 32), idx = (uint32_t)k;
		const bool head = in && (i == 0 || (uint32_t)(keys[i - 1] >> 32) != m);
		const bool tail = in && (i + 1 == n || (uint32_t)(keys[i + 1] >> 32) != m);
		uint32_t hmax; uint32_t hi = wtz_wg_incl_max(head ? i + 1 : 0u, tmp, &hmax);       /* position + 1 of the latest head at or before i */
		if(hi < hcarry) hi = hcarry;
		hcarry = hmax > hcarry ? hmax : hcarry;
		const uint32_t h0 = hi - 1, rank = in ? i - h0 : 0u, c = rank + 1;
		if(in){
			Z.sidx[o + i] = idx;
			Z.ok[o + idx] = (zcut > 255u) ? 1 : (rank < zcut ? 1 : 0);
		}
		const uint32_t fl = (tail && c < zcut) ? 1u : 0u;
		WTZ_WG_SYNC();
		uint32_t dtot; const uint32_t d = dcarry + wtz_wg_excl_scan(fl, tmp, &dtot);
		if(fl){ Z.dmer[o + d] = m; Z.dfirst[o + d] = h0; Z.dcnt[o + d] = (uint16_t)(c < 0xFFFFu ? c : 0xFFFFu); }
		dcarry += dtot;
		WTZ_WG_SYNC();
	}
	if(tid == 0) Z.dn[r] = dcarry;
}

/* the seed runs of a query's sampled k-mers, one WAVEFRONT per k-mer: koff[e] = run start << 16 | run length (wtz_kprobe), lanes take the run's entries side by
 * side (one coalesced read; four k-mers' first reads are issued together).  f(seed, k-mer index, k-mer length) is called for every entry that is neither the
 * query itself nor a read longer than 1.2 x the query (wtzmo.c:488-489).  Host emulation: one "lane". */
typedef struct { uint32_t nk; const uint64_t *koff; const wtz_kq_t *kq; const uint32_t *seeds; const uint32_t *rdlen; uint32_t pbid, thr, pblen_up; } wtz_cwg_walk_t;
template<typename F>
WTZ_HD void wtz_cwg_for_seeds(const wtz_cwg_walk_t &W, F &&f){
#if defined(__HIP_DEVICE_COMPILE__)
	/* 64 k-mers per step: every lane fetches one run descriptor, then the wave takes the runs four at a time with the descriptors read from the lanes'
	 * registers (v_readlane: no dependent load in front of the seed reads) */
	const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
	for(uint32_t base = wv * 64u; base < W.nk; base += nw * 64u){
		const uint32_t cnt = W.nk - base < 64u ? W.nk - base : 64u;
		const bool in = lane < cnt;
		const uint64_t myko = in ? W.koff[base + lane] : 0ull;
		const uint32_t myql = in ? W.kq[base + lane].qlen : 0u;
		const uint32_t my_lo = (uint32_t)myko, my_hi = (uint32_t)(myko >> 32);
		for(uint32_t u0 = 0; u0 < cnt; u0 += 4u){
			uint64_t ko[4]; uint32_t ql[4], s0[4];
			#pragma unroll
			for(uint32_t u = 0; u < 4u; u++){
				const uint32_t src = u0 + u < 64u ? u0 + u : 63u;       /* lanes >= cnt hold empty runs */
				ko[u] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)my_hi, (int)src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)my_lo, (int)src);
				ql[u] = (uint32_t)__builtin_amdgcn_readlane((int)myql, (int)src);
				if(u0 + u >= cnt) ko[u] = 0ull;
			}
			/* the first two chunks of 64 entries of all four runs are requested together; what a run has beyond them (hp-compressed 16-mers of a 10 Gbp read set: ~240 entries
			 * per k-mer) comes four chunks at a time - every load of a round is on its way before the first is used (round 5 fetched the later chunks one by one, each
			 * waiting out its own trip to HBM) */
			uint32_t s1[4];
			#pragma unroll
			for(uint32_t u = 0; u < 4u; u++){
				const uint32_t c = (uint32_t)(ko[u] & 0xFFFFu); const uint64_t o = ko[u] >> 16;
				s0[u] = lane < c ? W.seeds[o + lane] : 0u;
				s1[u] = lane + 64u < c ? W.seeds[o + 64u + lane] : 0u;
			}
			#pragma unroll
			for(uint32_t u = 0; u < 4u; u++){
				const uint32_t c = (uint32_t)(ko[u] & 0xFFFFu); const uint64_t o = ko[u] >> 16;
				auto take = [&](uint32_t sd){
					const bool drop = ((sd >> 1) == W.pbid) || (W.thr != 0xFFFFFFFFu ? (sd >> 1) < W.thr : W.rdlen[sd >> 1] > W.pblen_up);      /* wtzmo.c:488-489 */
					if(!drop) f(sd, base + u0 + u, ql[u]);
				};
				if(lane < c) take(s0[u]);
				if(lane + 64u < c) take(s1[u]);
				for(uint32_t k0 = 128u; k0 < c; k0 += 256u){
					uint32_t v[4];
					#pragma unroll
					for(uint32_t j = 0; j < 4u; j++){ const uint32_t k = k0 + j * 64u + lane; v[j] = k < c ? W.seeds[o + k] : 0u; }
					#pragma unroll
					for(uint32_t j = 0; j < 4u; j++){ if(k0 + j * 64u + lane < c) take(v[j]); }
				}
			}
		}
	}
#else
	for(uint32_t e = 0; e < W.nk; e++){
		const uint32_t c = (uint32_t)(W.koff[e] & 0xFFFFu); const uint64_t o = W.koff[e] >> 16; const uint32_t ql = W.kq[e].qlen;
		for(uint32_t k = 0; k < c; k++){
			const uint32_t sd = W.seeds[o + k];
			const bool drop = ((sd >> 1) == W.pbid) || (W.thr != 0xFFFFFFFFu ? (sd >> 1) < W.thr : W.rdlen[sd >> 1] > W.pblen_up);      /* wtzmo.c:488-489 */
			if(!drop) f(sd, e, ql);
		}
	}
#endif
}

struct wtz_kq2_f { uint64_t *mer; wtz_kq_t *kq; uint32_t n;
	WTZ_HDM void operator()(uint64_t m, uint32_t, uint32_t qo, uint32_t qe){ uint32_t l = qe - qo; if(l > 0xFFFFu) l = 0xFFFFu; mer[n] = m; kq[n].qoff = qo; kq[n].qlen = l; n++; } };

WTZ_HD void wtz_task_candidates_wg(uint32_t t, wtz_reads_t R, const uint32_t *qids, const wtz_params_t *P,
		const wtz_kslot_t *tab, uint64_t tmask, const uint32_t *seeds, wtz_pool_t *pool, uint64_t *cand_out, uint32_t *ncand_out, uint32_t stride,
		unsigned long long *algo_bytes, uint32_t *lds, const uint32_t *id_thr, uint32_t key_hi, uint64_t *gptr = NULL){
	const uint32_t tid = WTZ_WG_TID, nt = WTZ_WG_N;
	const uint32_t pbid = qids[t], L = R.rdlen[pbid];
	const uint32_t pblen_up = (uint32_t)(L * 1.2);                       /* double multiply, wtzmo.c:445 */
	const uint32_t thr = id_thr ? id_thr[t] : 0xFFFFFFFFu;              /* indexed reads in non-increasing length order: "longer than 1.2 x" is "id below thr" */
	uint32_t *tmp = lds;                                                 /* 64 words of scan / broadcast scratch (first: the sketch below takes everything behind it) */
	uint32_t *hist = lds + 64u;                                          /* WTZ_CWG_BINS counters / cursors, later the per-group sums */
	uint64_t *sbuf = (uint64_t*)(hist + WTZ_CWG_BINS);                   /* WTZ_CWG_CAP sort slots */
	uint32_t *ends = hist + WTZ_CWG_BINS + 2u * WTZ_CWG_CAP;             /* end / start of the tuple at each sorted position: 2 x CAP words */
	/* ---- A: the read's sampled k-mers ---- */
	unsigned long long pc = WTZ_CPROF_T(); (void)pc;
	if(tid == 0){
		const uint64_t pa = (uint64_t)(uintptr_t)wtz_pool_alloc(pool, (size_t)(L + 2) * 24);
		tmp[60] = (uint32_t)pa; tmp[61] = (uint32_t)(pa >> 32);
	}
	WTZ_WG_SYNC();
	uint8_t *mem = (uint8_t*)(uintptr_t)(((uint64_t)tmp[61] << 32) | tmp[60]);
	if(mem == NULL){ if(tid == 0) ncand_out[t] = 0xFFFFFFFFu; return; }
	uint64_t *kmer = (uint64_t*)mem; wtz_kq_t *kq = (wtz_kq_t*)(mem + (size_t)(L + 2) * 8); uint64_t *koff = (uint64_t*)(mem + (size_t)(L + 2) * 16);
	uint32_t nk;
	{
		const uint32_t PL = (L + nt - 1) / nt;
		const uint32_t jb = tid * PL, je = jb + PL;
		uint32_t cnt = 0;
		if(jb < L || (tid == 0 && L == 0)){ wtz_kcount_f fc; fc.n = 0; wtz_kmer_walk(R, pbid, P->ksize, P->hk, P->ksave, fc, jb, je); cnt = fc.n; }
		const uint32_t ex = wtz_wg_excl_scan(cnt, tmp, &nk);
		if(cnt){ wtz_kq2_f f; f.mer = kmer; f.kq = kq; f.n = ex; wtz_kmer_walk(R, pbid, P->ksize, P->hk, P->ksave, f, jb, je); }
	}
	WTZ_WG_SYNC();
	WTZ_CPROF_ADD(0, pc); WTZ_CPROF_CNT(8, 1); WTZ_CPROF_CNT(9, nk);
	/* ---- B: one hash probe per sampled k-mer (thread per k-mer: independent random reads); the seed runs themselves are then walked by a WAVEFRONT per
	 * k-mer (wtz_cwg_for_seeds): one coalesced read of the run instead of 64 lanes each stepping through a run of its own ---- */
	const uint32_t key_lo = thr != 0xFFFFFFFFu ? thr << 1 : 0u;
	uint32_t shift = 0; while(((key_hi > key_lo ? key_hi - key_lo : 1u) >> shift) > WTZ_CWG_BINS - 1u) shift++;
	const uint32_t kovl = P->kovl;
	unsigned long long my_T = 0; uint32_t T_all = 0;
	for(uint32_t e = tid; e < nk; e += nt){
		uint64_t o = 0; uint32_t c = 0;
		if(!wtz_kprobe(tab, tmask, kmer[e], &o, &c)){ o = 0; c = 0; }
		koff[e] = (o << 16) | c;
		my_T += c;
	}
	{   /* algorithmic bytes of this query */
		uint32_t tot_lo; const uint32_t dummy = wtz_wg_excl_scan((uint32_t)my_T, tmp, &tot_lo); (void)dummy;
		T_all = tot_lo;
		if(tid == 0){
			const unsigned long long bytes = (unsigned long long)L / 4 + 16ull * nk + 4ull * tot_lo;
			WTZ_CPROF_CNT(14, tot_lo);
#if defined(__HIP_DEVICE_COMPILE__)
			atomicAdd(algo_bytes, bytes);
#else
			*algo_bytes += bytes;
#endif
		}
	}
	/* ---- K: sketch.  Nearly every (read, strand) group is a chance hit of one or two k-mers and cannot reach -d; ol <= sum of the group's lengths, so a
	 * hashed table of that sum (16-bit counters in units of `sk_unit` bases, rounded up; a counter stops counting at the threshold) tells which tuples can
	 * belong to a group that does.  Collisions only add: no group that reaches -d is lost, and the groups that get through are folded exactly below. ---- */
	uint32_t *sk = hist;                                               /* WTZ_CWG_SKETCH 8-bit counters over bins + sort slots + interval arrays + 16 KB behind them */
	/* units of sk_unit bases, every tuple rounded UP (the sum of the round-ups is at least the round-up of the sum): a group whose lengths add up to -d holds
	 * at least sk_thr = ceil(-d / sk_unit) <= 255 units */
	const uint32_t sk_unit = kovl > 255u ? (kovl + 254u) / 255u : 1u, sk_thr = (kovl + sk_unit - 1u) / sk_unit;
	for(uint32_t i = tid; i < WTZ_CWG_SKETCH / 4u; i += nt) sk[i] = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	__threadfence_block();
#endif
	WTZ_WG_SYNC();
	const wtz_cwg_walk_t SW = { nk, koff, kq, seeds, R.rdlen, pbid, thr, pblen_up };
	wtz_cwg_for_seeds(SW, [&](uint32_t sd, uint32_t, uint32_t qlen){
		const uint32_t l = qlen < kovl ? qlen : kovl;
		wtz_cwg_sk_add(sk, wtz_cwg_sk_hash(sd), (l + sk_unit - 1u) / sk_unit, sk_thr);
	});
	WTZ_WG_SYNC();
	WTZ_CPROF_ADD(1, pc);
	/* ---- H: the tuples the sketch lets through, listed once (in any order): the scatter below reads them back instead of walking all the runs a third time ---- */
	if(tid == 0){
		const uint64_t pa = (uint64_t)(uintptr_t)wtz_pool_alloc(pool, ((size_t)T_all + 2) * 8);
		tmp[60] = (uint32_t)pa; tmp[61] = (uint32_t)(pa >> 32); tmp[59] = 0;
	}
	WTZ_WG_SYNC();
	uint64_t *lst_t = (uint64_t*)(uintptr_t)(((uint64_t)tmp[61] << 32) | tmp[60]);
	if(lst_t == NULL){ if(tid == 0) ncand_out[t] = 0xFFFFFFFFu; return; }
	/* a listed tuple: key << 32 | k-mer index << 8 | sketch units (a read has fewer than 2^24 bases; the index stays above the units, so that the sort below still orders
	 * a group's tuples by query offset; the units ride along so that the second level needs no gather through the k-mer table: a dependent load per tuple was most of its time) */
	wtz_cwg_for_seeds(SW, [&](uint32_t sd, uint32_t e0, uint32_t qlen){
		const uint32_t l0 = qlen < kovl ? qlen : kovl;
		const uint32_t e = ((e0 & 0xFFFFFFu) << 8) | ((l0 + sk_unit - 1u) / sk_unit);
		const bool ps = wtz_cwg_sk_pass(sk, wtz_cwg_sk_hash(sd), sk_thr);
#if defined(__HIP_DEVICE_COMPILE__)
		/* one LDS atomic per wavefront instruction, not per tuple (512 threads on one counter): the lanes that pass take consecutive slots behind the leader's */
		const unsigned long long pm = __ballot(ps);
		if(pm){
			const uint32_t ldr = (uint32_t)__builtin_ctzll(pm);
			uint32_t at = 0;
			if((threadIdx.x & 63u) == ldr) at = atomicAdd(&tmp[59], (uint32_t)__popcll(pm));
			at = (uint32_t)__shfl((int)at, (int)ldr, 64);
			if(ps) lst_t[at + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u))] = ((uint64_t)sd << 32) | e;
		}
#else
		if(ps) lst_t[WTZ_LDS_ADD32(&tmp[59], 1u)] = ((uint64_t)sd << 32) | e;
#endif
	});
#if defined(__HIP_DEVICE_COMPILE__)
	__threadfence_block();
#endif
	WTZ_WG_SYNC();
	uint32_t n_listed = tmp[59];
	WTZ_WG_SYNC();
	/* ---- second level (round 6): where many counters reached the threshold only because ~30 groups share each (a long query at the configs[3] shape: 37 % of 460 000
	 * tuples listed for 41 groups that reach -d), the LISTED tuples are counted again under another hash, in as many parts of the key space as keep the counters sparse (the longest
	 * queries list a million tuples), and the list is compacted to what passes that too.  Exact: a group that reaches -d brought all its tuples through the first level (they share its counter), so its second counter holds its whole
	 * sum as well. ---- */
	if(n_listed > WTZ_CWG_L2_MIN){
		/* as many parts as keep the average counter near a third of the threshold (a listed tuple is ~12 units); a group's tuples share its key, hence its part */
		const uint32_t parts = (uint32_t)(((uint64_t)n_listed * 12u + WTZ_CWG_L2_PART_UNITS - 1u) / WTZ_CWG_L2_PART_UNITS);
		const uint64_t DEAD = ~0ull;
		for(uint32_t part = 0; part < parts; part++){
			for(uint32_t i = tid; i < WTZ_CWG_SKETCH / 4u; i += nt) sk[i] = 0;
#if defined(__HIP_DEVICE_COMPILE__)
			__threadfence_block();
#endif
			WTZ_WG_SYNC();
			for(uint32_t i0 = 0; i0 < n_listed; i0 += 4u * nt){      /* four loads on their way per thread before the first is used */
				uint64_t w4[4];
				#pragma unroll
				for(uint32_t j = 0; j < 4u; j++){ const uint32_t i = i0 + j * nt + tid; w4[j] = i < n_listed ? lst_t[i] : DEAD; }
				#pragma unroll
				for(uint32_t j = 0; j < 4u; j++){
					if(w4[j] == DEAD) continue;
					const uint32_t sd2 = (uint32_t)(w4[j] >> 32);
					if(parts > 1u && ((wtz_cwg_sk_hash3(sd2) * parts) >> 16) != part) continue;
					wtz_cwg_sk_add(sk, wtz_cwg_sk_hash2(sd2), ((uint32_t)w4[j]) & 255u, sk_thr);
				}
			}
			WTZ_WG_SYNC();
			for(uint32_t i0 = 0; i0 < n_listed; i0 += 4u * nt){      /* every thread marks the slots it read itself */
				uint64_t w4[4];
				#pragma unroll
				for(uint32_t j = 0; j < 4u; j++){ const uint32_t i = i0 + j * nt + tid; w4[j] = i < n_listed ? lst_t[i] : DEAD; }
				#pragma unroll
				for(uint32_t j = 0; j < 4u; j++){
					if(w4[j] == DEAD) continue;
					const uint32_t sd2 = (uint32_t)(w4[j] >> 32);
					if(parts > 1u && ((wtz_cwg_sk_hash3(sd2) * parts) >> 16) != part) continue;
					if(!wtz_cwg_sk_pass(sk, wtz_cwg_sk_hash2(sd2), sk_thr)) lst_t[i0 + j * nt + tid] = DEAD;
				}
			}
			WTZ_WG_SYNC();
		}
		/* in-place compaction (any order), a round of 4 x nt tuples at a time: a round is read before anything of it is written, and it writes at or in front of its own first slot */
		uint32_t kept = 0;
		for(uint32_t i0 = 0; i0 < n_listed; i0 += 4u * nt){
			uint64_t w[4]; uint32_t keep = 0;
			#pragma unroll
			for(uint32_t j = 0; j < 4u; j++){
				const uint32_t i = i0 + j * nt + tid;
				w[j] = DEAD;
				if(i < n_listed) w[j] = lst_t[i];
				if(w[j] != DEAD) keep |= 1u << j;
			}
			uint32_t tot; uint32_t at = kept + wtz_wg_excl_scan((uint32_t)__builtin_popcount(keep), tmp, &tot);      /* (its barriers also separate this round's reads from its writes) */
			#pragma unroll
			for(uint32_t j = 0; j < 4u; j++) if((keep >> j) & 1u) lst_t[at++] = w[j];
			kept += tot;
			WTZ_WG_SYNC();
		}
		n_listed = kept;
#if defined(__HIP_DEVICE_COMPILE__)
		__threadfence_block();
#endif
		WTZ_WG_SYNC();
	}
	/* the sketch is dead: its first 16 KB become the histogram of the listed tuples over the key bins */
	for(uint32_t i = tid; i < WTZ_CWG_BINS; i += nt) hist[i] = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	__threadfence_block();
#endif
	WTZ_WG_SYNC();
	for(uint32_t i = tid; i < n_listed; i += nt) WTZ_LDS_ADD32(&hist[((uint32_t)(lst_t[i] >> 32) - key_lo) >> shift], 1u);
#if defined(__HIP_DEVICE_COMPILE__)
	__threadfence_block();
#endif
	WTZ_WG_SYNC();
	/* ---- S: bin offsets, buckets, scatter ---- */
	uint32_t Tk = 0;
	{
		const uint32_t per = WTZ_CWG_BINS / nt ? WTZ_CWG_BINS / nt : WTZ_CWG_BINS;      /* bins per thread (host emulation: all of them) */
		uint32_t loc = 0;
		for(uint32_t i = 0; i < per; i++) if(tid * per + i < WTZ_CWG_BINS) loc += hist[tid * per + i];
		uint32_t run = wtz_wg_excl_scan(loc, tmp, &Tk);
		WTZ_WG_SYNC();
		for(uint32_t i = 0; i < per; i++) if(tid * per + i < WTZ_CWG_BINS){ const uint32_t v = hist[tid * per + i]; hist[tid * per + i] = run; run += v; }
	}
	WTZ_WG_SYNC();
	if(tid == 0){
		const uint64_t pa = Tk ? (uint64_t)(uintptr_t)wtz_pool_alloc(pool, ((size_t)Tk + 2) * 8) : 1ull;
		tmp[60] = (uint32_t)pa; tmp[61] = (uint32_t)(pa >> 32);
	}
	WTZ_WG_SYNC();
	uint64_t *tup = (uint64_t*)(uintptr_t)(((uint64_t)tmp[61] << 32) | tmp[60]);
	if(tup == NULL){ if(tid == 0) ncand_out[t] = 0xFFFFFFFFu; return; }
	WTZ_CPROF_ADD(2, pc); WTZ_CPROF_CNT(10, Tk);
	/* groups that reach -d, in key order (at most one per tuple): they take the place of the tuple LIST, which is dead once the scatter below has read it (round 5: a third
	 * of the scratch of a query - 26 MB at the configs[3] shape, which is what caps the batch there) */
	uint64_t *grp = lst_t;
	const uint32_t grp_cap = T_all + 2;
	for(uint32_t i = tid; i < Tk; i += nt){ const uint64_t w = lst_t[i]; tup[WTZ_LDS_ADD32(&hist[((uint32_t)(w >> 32) - key_lo) >> shift], 1u)] = w; }
#if defined(__HIP_DEVICE_COMPILE__)
	__threadfence_block();
#endif
	WTZ_WG_SYNC();
	WTZ_CPROF_ADD(3, pc);
	/* ---- P: per bucket: sort, then the union length of every group ---- */
	uint32_t ng = 0; int over = 0;
	/* buckets = consecutive bins while their tuples fit the sort slots (a bin above CAP is a bucket of its own); after the scatter a bin's cursor
	 * stands at the start of the next bin, so bin b covers [hist[b - 1], hist[b]) and hist[] is non-decreasing: the end of a bucket is a binary search
	 * that every thread does for itself (LDS broadcasts; thread 0 stepping through the bins was a twelfth of the kernel) */
	for(uint32_t bin0 = 0; bin0 < WTZ_CWG_BINS; ){
		const uint32_t t0 = bin0 ? hist[bin0 - 1] : 0u;
		uint32_t bin1;
		{
			uint32_t lo = bin0 + 1, hi = WTZ_CWG_BINS;      /* the first b in (bin0, BINS) with hist[b] - t0 > CAP, or BINS */
			while(lo < hi){ const uint32_t mid = lo + (hi - lo) / 2u; if(hist[mid] - t0 > WTZ_CWG_CAP) hi = mid; else lo = mid + 1; }
			bin1 = lo;
		}
		const uint32_t t1 = hist[bin1 - 1];
		bin0 = bin1;
		const uint32_t n = t1 - t0;
		WTZ_CPROF_ADD(4, pc);
		if(n == 0) continue;
		WTZ_CPROF_CNT(11, 1);
		uint64_t *srt = sbuf; uint32_t *en = ends, *qo = ends + WTZ_CWG_CAP;
		uint32_t np = 64; while(np < n) np <<= 1;
		const bool in_lds = n <= WTZ_CWG_CAP;
		WTZ_WG_SYNC();                                      /* the previous bucket's readers are done with the sort slots */
		if(!in_lds){
			/* one bin alone holds more than CAP tuples (a read that shares thousands of k-mers with the query: duplicates, repeats): same steps on
			 * arrays in the pool */
			if(tid == 0){
				const uint64_t pa = (uint64_t)(uintptr_t)wtz_pool_alloc(pool, (size_t)np * 16);
				tmp[60] = (uint32_t)pa; tmp[61] = (uint32_t)(pa >> 32);
			}
			WTZ_WG_SYNC();
			srt = (uint64_t*)(uintptr_t)(((uint64_t)tmp[61] << 32) | tmp[60]);
			if(srt == NULL){ over = 1; break; }
			en = (uint32_t*)(srt + np); qo = en + np;
		}
		for(uint32_t i = tid; i < np; i += nt) srt[i] = i < n ? tup[t0 + i] : ~0ull;
#if defined(__HIP_DEVICE_COMPILE__)
		__threadfence_block();
#endif
		WTZ_WG_SYNC();
		wtz_wg_sort_u64(srt, np);
		WTZ_CPROF_ADD(5, pc);
		/* query interval of every tuple in sorted order: the fold below then touches LDS only */
		for(uint32_t i = tid; i < n; i += nt){ const wtz_kq_t q = kq[(uint32_t)srt[i] >> 8]; qo[i] = q.qoff; en[i] = q.qoff + q.qlen; }
#if defined(__HIP_DEVICE_COMPILE__)
		__threadfence_block();
#endif
		WTZ_WG_SYNC();
		if(in_lds){
			/* the union-length recurrence of wtzmo.c:558-560 without a walk: after a group's first tuple `lst` is the end of the tuple before, so the addend of
			 * tuple i is  en[i] - (qo[i] >= en[i-1] ? qo[i] : en[i-1])  (first tuple: en - qo; u32 arithmetic as the reference's) and ol is a segmented sum.  Two
			 * workgroup scans over the sorted bucket: S = running sum of the addends (kept where qo was), H = position + 1 of the latest group start (kept in the
			 * low word of the sort slot: the k-mer index there is dead); the last tuple of a group then has ol = S[i] - S[H[i] - 2]. */
			uint32_t carry_s = 0, carry_h = 0;
			for(uint32_t i0 = 0; i0 < n; i0 += nt){
				const uint32_t i = i0 + tid;
				uint32_t add = 0, hd = 0;
				if(i < n){
					const uint32_t key = (uint32_t)(srt[i] >> 32), q0 = qo[i], e1 = en[i];
					if(i == 0 || (uint32_t)(srt[i - 1] >> 32) != key){ add = e1 - q0; hd = i + 1; }
					else { const uint32_t lst = en[i - 1]; add = q0 >= lst ? e1 - q0 : e1 - lst; }
				}
				uint32_t tot_s, tot_h;
				const uint32_t ex = wtz_wg_excl_scan(add, tmp, &tot_s);
				const uint32_t hm = wtz_wg_incl_max(hd, tmp, &tot_h);
				if(i < n){
					qo[i] = carry_s + ex + add;
					const uint32_t hh = hm > carry_h ? hm : carry_h;
					srt[i] = (srt[i] & 0xFFFFFFFF00000000ull) | hh;
				}
				carry_s += tot_s; if(tot_h > carry_h) carry_h = tot_h;
			}
#if defined(__HIP_DEVICE_COMPILE__)
			__threadfence_block();
#endif
			WTZ_WG_SYNC();
			for(uint32_t i0 = 0; i0 < n; i0 += nt){
				const uint32_t i = i0 + tid;
				uint32_t keep = 0; uint64_t g = 0;
				if(i < n){
					const uint32_t key = (uint32_t)(srt[i] >> 32);
					if(i + 1 == n || (uint32_t)(srt[i + 1] >> 32) != key){
						const uint32_t hh = (uint32_t)srt[i];                 /* >= 1: tuple 0 starts a group */
						const uint32_t ol = qo[i] - (hh >= 2u ? qo[hh - 2u] : 0u);
						if(ol >= kovl){ keep = 1; g = ((uint64_t)key << 32) | ol; }
					}
				}
				uint32_t chunk; const uint32_t ex = wtz_wg_excl_scan(keep, tmp, &chunk);
				if(keep){ if(ng + ex < grp_cap) grp[ng + ex] = g; else over = 1; }
				ng += chunk;
			}
		} else {
			/* the thread at a group's first tuple folds the group (wtzmo.c:558-560) */
			for(uint32_t i0 = 0; i0 < n; i0 += nt){
				const uint32_t i = i0 + tid;
				uint32_t keep = 0; uint64_t g = 0;
				if(i < n){
					const uint32_t key = (uint32_t)(srt[i] >> 32);
					if(i == 0 || (uint32_t)(srt[i - 1] >> 32) != key){
						uint32_t ol = 0, lst = 0;
						for(uint32_t r = i; r < n && (uint32_t)(srt[r] >> 32) == key; r++){
							const uint32_t q0 = qo[r], e1 = en[r];
							if(q0 >= lst) ol += e1 - q0; else ol += e1 - lst;       /* wtzmo.c:558-559: len, or off + len - lst (u32 arithmetic) */
							lst = e1;
						}
						if(ol >= kovl){ keep = 1; g = ((uint64_t)key << 32) | ol; }
					}
				}
				uint32_t chunk; const uint32_t ex = wtz_wg_excl_scan(keep, tmp, &chunk);
				if(keep){ if(ng + ex < grp_cap) grp[ng + ex] = g; else over = 1; }
				ng += chunk;
			}
		}
		WTZ_CPROF_ADD(6, pc);
	}
#if defined(__HIP_DEVICE_COMPILE__)
	over = __syncthreads_or(over);
#endif
	if(over || ng > grp_cap){ if(tid == 0) ncand_out[t] = 0xFFFFFFFFu; return; }
#if defined(__HIP_DEVICE_COMPILE__)
	__threadfence_block();
#endif
	WTZ_WG_SYNC();
	/* ---- F: strand merge + candidate heap (wtzmo.c:516-571), one thread by definition.  Round 4: the heap (<= -A + 1 entries) and the groups it consumes
	 * sit in LDS - every sift level on a heap in the pool was a dependent L2 round trip, and with -A 1000 (the dmo pipeline) most groups are pushed: the seed
	 * lookup of a configs[2] dmo step took 460 ms against 87 ms with -A 500.  The sort / partition arrays are dead here. ---- */
	if(gptr){ if(tid == 0){ gptr[t] = (uint64_t)(uintptr_t)grp; ncand_out[t] = ng; } }      /* sharded index: the groups go to the caller, who joins the shards (wtz_cand_tail_host) */
	else if(P->ncand + 1u <= WTZ_CWG_CAP){
		uint64_t *hp = sbuf, *gb = (uint64_t*)ends;             /* WTZ_CWG_CAP words each */
		uint64_t *row = cand_out + (size_t)t * stride;
		const uint32_t hn0 = ncand_out[t];                      /* heap carried across index parts (-G), 0 otherwise */
		for(uint32_t i = tid; i < hn0; i += nt) hp[i] = row[i];
		uint64_t x1 = WTZ_CAND_NONE; uint32_t n = hn0;          /* live on thread 0 */
		for(uint32_t c0 = 0; c0 < ng; c0 += WTZ_CWG_CAP){
			const uint32_t m = ng - c0 < WTZ_CWG_CAP ? ng - c0 : WTZ_CWG_CAP;
			WTZ_WG_SYNC();
			for(uint32_t i = tid; i < m; i += nt) gb[i] = grp[c0 + i];
			WTZ_WG_SYNC();
			if(tid == 0) wtz_cand_tail_step(gb, m, kovl, P->ncand, hp, n, x1);
		}
		WTZ_WG_SYNC();
		if(tid == 0){ wtz_cand_tail_finish(P->ncand, hp, n, x1); tmp[0] = n; }
		WTZ_WG_SYNC();
		n = tmp[0];
		for(uint32_t i = tid; i < n; i += nt) row[i] = hp[i];
		if(tid == 0) ncand_out[t] = n;
		WTZ_CPROF_ADD(7, pc); WTZ_CPROF_CNT(12, ng); WTZ_CPROF_CNT(13, n);
	} else if(tid == 0){
		uint32_t hn = ncand_out[t];
		wtz_cand_tail(grp, ng, kovl, P->ncand, cand_out + (size_t)t * stride, &hn);
		ncand_out[t] = hn;
	}
}

struct wtz_kq_f { uint64_t *mer; uint32_t *qoff, *qlen; uint32_t n;
	WTZ_HDM void operator()(uint64_t m, uint32_t, uint32_t qo, uint32_t qe){ uint32_t l = qe - qo; if(l > 0xFFFFu) l = 0xFFFFu; mer[n] = m; qoff[n] = qo; qlen[n] = l; n++; } };

/*
 * task (wave-cooperative): candidates of query qids[t]; cand_out row stride = ncand + 1.
 *   A  lane 0 walks the read once and lists its sampled k-mers (the hp-compressed walk is a serial recurrence)
 *   B  all lanes probe the hash (one 16-byte slot load per k-mer) and lay out the seed runs with an exclusive scan
 *   C  all lanes expand (read<<1|strand, query offset, length) tuples, dropping self hits and reads longer than 1.2x
 *      (wtzmo.c:488-489); tuple sequence numbers follow query-offset order
 *   D  wave-wide bitonic sort of (key<<32 | sequence): per (read,strand) group the tuples stay in query-offset order,
 *      which is the order the reference's k-way heap merge delivers them in (wtzmo.c:44-57)
 *   E  all lanes: group heads -> union length `ol` of each group (wtzmo.c:558-560), compacted in key order
 *   F  lane 0 replays the strand merge + candidate heap with its quirks (wtzmo.c:516-571)
 */
template<bool STREAM = false>
WTZ_HD void wtz_task_candidates(uint32_t t, wtz_reads_t R, const uint32_t *qids, const wtz_params_t *P,
		const wtz_kslot_t *tab, uint64_t tmask, const uint32_t *seeds, wtz_pool_t *pool, uint64_t *cand_out, uint32_t *ncand_out, uint32_t stride,
		unsigned long long *algo_bytes, uint64_t *lds, uint32_t lds_words, const uint32_t *id_thr = NULL){
	const uint32_t pbid = qids[t], lane = WTZ_LANE;
	const uint32_t L = R.rdlen[pbid];
	const uint32_t pblen_up = (uint32_t)(L * 1.2);                       /* double multiply, wtzmo.c:445 */
	/* ---- A ---- */
	const unsigned long long pcA = WTZ_PROF_T(); (void)pcA;
	uint64_t pa = 0; uint32_t nk = 0;
	if(lane == 0) pa = (uint64_t)(uintptr_t)wtz_pool_alloc(pool, (size_t)(L + 2) * 16 + (size_t)(L + 2) * 12);
	pa = wtz_coop_bcast64(pa);
	if(pa == 0){ if(lane == 0) ncand_out[t] = 0xFFFFFFFFu; return; }
	uint8_t *mem = (uint8_t*)(uintptr_t)pa;
	{   /* every lane walks one piece of the read (exact restart: wtz_walk_warm_start), counts, then writes at its offset */
		const uint32_t PL = (L + WTZ_NLANES - 1) / WTZ_NLANES;
		const uint32_t jb = lane * PL, je = jb + PL;
		uint32_t cnt = 0;
		if(jb < L || (lane == 0 && L == 0)){ wtz_kcount_f fc; fc.n = 0; wtz_kmer_walk(R, pbid, P->ksize, P->hk, P->ksave, fc, jb, je); cnt = fc.n; }
		uint32_t tot; const uint32_t ex = wtz_coop_excl_scan(cnt, &tot);
		nk = tot;
		if(cnt){
			wtz_kq_f f; f.mer = (uint64_t*)mem; f.qoff = (uint32_t*)(mem + (size_t)(L + 2) * 8); f.qlen = f.qoff + (L + 2); f.n = ex;
			wtz_kmer_walk(R, pbid, P->ksize, P->hk, P->ksave, f, jb, je);
		}
	}
	WTZ_WAVE_SYNC();
	const uint64_t *kmer = (const uint64_t*)mem; const uint32_t *kqoff = (const uint32_t*)(mem + (size_t)(L + 2) * 8), *kqlen = kqoff + (L + 2);
	uint64_t *koff = (uint64_t*)(mem + (size_t)(L + 2) * 16);               /* seed run start per k-mer */
	uint32_t *ktoff = (uint32_t*)(koff + (L + 2));                           /* tuple offset per k-mer (cnt kept in the high part of koff) */
	WTZ_PROF_ADD(24, pcA);
	const unsigned long long pcB = WTZ_PROF_T(); (void)pcB;
	/* ---- B ---- */
	uint32_t T = 0;
	for(uint32_t e0 = 0; e0 < nk; e0 += WTZ_NLANES){
		const uint32_t e = e0 + lane;
		uint64_t o = 0; uint32_t c = 0;
		if(e < nk){ if(!wtz_kprobe(tab, tmask, kmer[e], &o, &c)){ o = 0; c = 0; } }
		uint32_t chunk; const uint32_t ex = wtz_coop_excl_scan(c, &chunk);
		if(e < nk){ koff[e] = (o << 16) | c; ktoff[e] = T + ex; }
		T += chunk;
	}
	if(lane == 0){
		const unsigned long long bytes = (unsigned long long)L / 4 + 16ull * nk + 4ull * T;      /* SURVEY 8d */
#if defined(__HIP_DEVICE_COMPILE__)
		atomicAdd(algo_bytes, bytes);
#else
		*algo_bytes += bytes;
#endif
	}
	WTZ_PROF_ADD(25, pcB); WTZ_PROF_CNT(30, T); WTZ_PROF_CNT(31, nk);
	const unsigned long long pcC = WTZ_PROF_T(); (void)pcC;
#if defined(__HIP_DEVICE_COMPILE__)
	/* the streaming form (no tuples, no sort) when the launch gave the wave its table; a query with too many groups falls through.
	 * A separate instantiation: the sorting form keeps its register budget (and occupancy) when the streaming form is not asked for */
	if(STREAM && lds && lds_words * 8u >= WTZ_CAND_STREAM_LDS_BYTES(P->ncand)){
		WTZ_WAVE_SYNC();
		if(wtz_cand_stream(t, R, pbid, pblen_up, P, seeds, nk, koff, kqoff, kqlen, cand_out, ncand_out, stride, (uint32_t*)lds, id_thr ? id_thr[t] : 0xFFFFFFFFu)){ WTZ_PROF_ADD(26, pcC); return; }
		WTZ_WAVE_SYNC();
		/* the sorting form below wants a power-of-two LDS window, and wants it as a compile-time constant (constant strides in the
		 * bitonic network): the sketch + table alone are 32 KB */
		lds_words = (WTZ_CAND_SKETCH + 3u * WTZ_CAND_TAB) * 4u / 8u;
	}
#endif
	/* ---- C ---- */
	uint32_t np = 64; while(np < T) np <<= 1;
	pa = 0;
	if(lane == 0) pa = (uint64_t)(uintptr_t)wtz_pool_alloc(pool, (size_t)np * 8 + (size_t)(T + 2) * 8 + (size_t)(T + 2) * 8);
	pa = wtz_coop_bcast64(pa);
	if(pa == 0){ if(lane == 0) ncand_out[t] = 0xFFFFFFFFu; return; }
	uint64_t *tup = (uint64_t*)(uintptr_t)pa, *tv = tup + np, *grp = tv + (T + 2);
	for(uint32_t i = T + lane; i < np; i += WTZ_NLANES) tup[i] = ~0ull;
#if defined(__HIP_DEVICE_COMPILE__)
	__threadfence_block();
#endif
	for(uint32_t e = lane; e < nk; e += WTZ_NLANES){
		const uint32_t c = (uint32_t)(koff[e] & 0xFFFFu); const uint64_t o = koff[e] >> 16;
		const uint32_t base = ktoff[e];
		const uint64_t v = ((uint64_t)kqoff[e] << 16) | kqlen[e];
		for(uint32_t k = 0; k < c; k++){
			const uint32_t sd = seeds[o + k];
			const bool drop = ((sd >> 1) == pbid) || (R.rdlen[sd >> 1] > pblen_up);          /* wtzmo.c:488-489 */
			tup[base + k] = drop ? ~0ull : (((uint64_t)sd << 32) | (base + k));
			tv[base + k] = v;
		}
	}
	WTZ_PROF_ADD(26, pcC);
	const unsigned long long pcD = WTZ_PROF_T(); (void)pcD;
	/* ---- D ---- */
	wtz_coop_sort_u64_windowed(tup, np, lds, lds_words);
	WTZ_PROF_ADD(27, pcD);
	const unsigned long long pcE = WTZ_PROF_T(); (void)pcE;
	/* ---- E ---- */
	uint32_t ng = 0;
	for(uint32_t i0 = 0; i0 < T; i0 += WTZ_NLANES){
		const uint32_t i = i0 + lane;
		uint32_t head = 0; uint64_t g = 0;
		if(i < T && tup[i] != ~0ull){
			const uint32_t key = (uint32_t)(tup[i] >> 32);
			if(i == 0 || (uint32_t)(tup[i - 1] >> 32) != key){
				head = 1;
				uint32_t ol = 0, lst = 0;
				for(uint32_t r = i; r < T && tup[r] != ~0ull && (uint32_t)(tup[r] >> 32) == key; r++){
					const uint64_t v = tv[(uint32_t)tup[r]];
					const uint32_t qo = (uint32_t)(v >> 16), ql = (uint32_t)(v & 0xFFFFu);
					if(qo >= lst) ol += ql; else ol += qo + ql - lst;                           /* wtzmo.c:558-559 */
					lst = qo + ql;
				}
				g = ((uint64_t)key << 32) | ol;
			}
		}
		uint32_t chunk; const uint32_t ex = wtz_coop_excl_scan(head, &chunk);
		if(head) grp[ng + ex] = g;
		ng += chunk;
	}
#if defined(__HIP_DEVICE_COMPILE__)
	__threadfence_block();
#endif
	WTZ_PROF_ADD(28, pcE);
	const unsigned long long pcF = WTZ_PROF_T(); (void)pcF;
	/* ---- F ---- */
	if(lane == 0){
		uint32_t hn = ncand_out[t];                 /* heap carried across index parts (-G), 0 otherwise */
		wtz_cand_tail(grp, ng, P->kovl, P->ncand, cand_out + (size_t)t * stride, &hn);
		ncand_out[t] = hn;
	}
	WTZ_PROF_ADD(29, pcF);
}

#endif
