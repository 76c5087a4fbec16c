/* seedstats.c - analysis aid (not product, not oracle): per-query statistics of the seed lookup (A3, wtzmo.c:433-573) on a FASTA set:
 * number of sampled k-mers, seed-run tuples, DISTINCT (read,strand) groups, groups whose length sum reaches -d, groups with ol >= -d.
 * Used to size the LDS group tables of K_candidates.  Own code; walks reads exactly like the index (hp-compressed canonical k-mers,
 * Jenkins subsample).  usage: seedstats reads.fa K nq   (K = k-mer frequency cutoff, nq = number of longest reads used as queries) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
typedef struct { char *s; uint32_t len; uint32_t id; } rd_t;
static uint32_t jenkins32(uint32_t key){ key += (key << 12); key ^= (key >> 22); key += (key << 4); key ^= (key >> 9); key += (key << 10); key ^= (key >> 2); key += (key << 7); key ^= (key >> 12); return key; }
static uint64_t revcomp(uint64_t x, unsigned k){ x = ~x; x = ((x & 0x3333333333333333ULL) << 2) | ((x & 0xCCCCCCCCCCCCCCCCULL) >> 2); x = ((x & 0x0F0F0F0F0F0F0F0FULL) << 4) | ((x & 0xF0F0F0F0F0F0F0F0ULL) >> 4);
	x = ((x & 0x00FF00FF00FF00FFULL) << 8) | ((x & 0xFF00FF00FF00FF00ULL) >> 8); x = ((x & 0x0000FFFF0000FFFFULL) << 16) | ((x & 0xFFFF0000FFFF0000ULL) >> 16); x = (x << 32) | (x >> 32); return x >> (64 - (k << 1)); }
static int cmp_len(const void *a, const void *b){ const rd_t *x = a, *y = b; return x->len < y->len ? 1 : (x->len > y->len ? -1 : 0); }
typedef struct { uint64_t mer; uint32_t sd, qoff; uint32_t len, q; } occ_t;
/* open addressing set of query k-mers */
static uint64_t *hs; static uint64_t hmask;
static uint64_t mix64(uint64_t x){ x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
static void hs_put(uint64_t m){ uint64_t h = mix64(m) & hmask; while(hs[h] != ~0ULL && hs[h] != m) h = (h + 1) & hmask; hs[h] = m; }
static int hs_has(uint64_t m){ uint64_t h = mix64(m) & hmask; while(hs[h] != ~0ULL){ if(hs[h] == m) return 1; h = (h + 1) & hmask; } return 0; }
typedef struct { uint64_t mer; uint32_t sd; } ko_t;
static int cmp_ko(const void *a, const void *b){ const ko_t *x = a, *y = b; if(x->mer != y->mer) return x->mer < y->mer ? -1 : 1; return x->sd < y->sd ? -1 : (x->sd > y->sd); }
typedef struct { uint32_t sd, qoff, len; } tup_t;
static int cmp_tup(const void *a, const void *b){ const tup_t *x = a, *y = b; if(x->sd != y->sd) return x->sd < y->sd ? -1 : 1; return x->qoff < y->qoff ? -1 : (x->qoff > y->qoff); }
#define WALK(R, ...) do { uint64_t mask = 0xFFFFFFFFFFFFFFFFULL >> ((32 - 16) << 1), kmer = 0; uint32_t i = 0, b = 4, ring[32]; \
	for(uint32_t j = 0; j < (R)->len; j++){ uint32_t c; switch((R)->s[j]){ case 'A': case 'a': c = 0; break; case 'C': case 'c': c = 1; break; case 'G': case 'g': c = 2; break; default: c = 3; } \
		if(c == b) continue; b = c; i++; ring[(i - 1) & 31] = j; kmer = ((kmer << 2) | b) & mask; if(i < 16) continue; \
		uint64_t rev = revcomp(kmer, 16); if(rev == kmer) continue; uint32_t dir = rev > kmer ? 0 : 1; uint64_t mer = rev > kmer ? kmer : rev; \
		if(jenkins32((uint32_t)mer) % 4096u >= 1024u) continue; uint32_t qoff = ring[(i - 16) & 31], qend = j + 1; (void)dir; (void)qoff; (void)qend; __VA_ARGS__ } } while(0)
int main(int argc, char **argv){
	if(argc < 4) return 1;
	FILE *f = fopen(argv[1], "r"); uint32_t K = atoi(argv[2]), nq = atoi(argv[3]), kovl = 300;
	size_t cap = 1 << 20, n = 0; rd_t *R = malloc(cap * sizeof(rd_t)); char *line = NULL; size_t lc = 0; ssize_t l;
	while((l = getline(&line, &lc, f)) > 0){ if(line[0] == '>') continue; while(l && (line[l-1] == '\n' || line[l-1] == '\r')) l--; if(n == cap){ cap *= 2; R = realloc(R, cap * sizeof(rd_t)); } R[n].s = malloc(l + 1); memcpy(R[n].s, line, l); R[n].len = l; n++; }
	qsort(R, n, sizeof(rd_t), cmp_len); for(size_t i = 0; i < n; i++) R[i].id = i;
	fprintf(stderr, "%zu reads, longest %u\n", n, R[0].len);
	/* queries: nq reads spread over the first 15%% of the ids (what is queried before masking ends it) */
	uint32_t *Q = malloc(nq * 4); for(uint32_t k = 0; k < nq; k++) Q[k] = (uint32_t)((uint64_t)k * (n * 15 / 100) / nq);
	hmask = (1ull << 24) - 1; hs = malloc((hmask + 1) * 8); memset(hs, 0xFF, (hmask + 1) * 8);
	for(uint32_t k = 0; k < nq; k++){ rd_t *r = &R[Q[k]]; WALK(r, hs_put(mer);); }
	size_t kc = 1 << 24, kn = 0; ko_t *KO = malloc(kc * sizeof(ko_t));
	for(size_t i = 0; i < n; i++){ rd_t *r = &R[i]; WALK(r, if(hs_has(mer)){ if(kn == kc){ kc *= 2; KO = realloc(KO, kc * sizeof(ko_t)); } KO[kn].mer = mer; KO[kn].sd = (r->id << 1) | dir; kn++; }); }
	qsort(KO, kn, sizeof(ko_t), cmp_ko);
	fprintf(stderr, "%zu occurrences of the queries' k-mers\n", kn);
	printf("#qid\tlen\tnk\tnk_hit\tT\tT_kept\tG\tG_sum>=d\tG_ol>=d\tmaxrun\n");
	size_t tc = 1 << 22; tup_t *T = malloc(tc * sizeof(tup_t));
	for(uint32_t k = 0; k < nq; k++){
		rd_t *r = &R[Q[k]]; size_t tn = 0, Tall = 0; uint32_t nk = 0, nkh = 0, maxrun = 0; uint32_t up = (uint32_t)(r->len * 1.2);
		WALK(r, { nk++; size_t lo = 0; size_t hi = kn; while(lo < hi){ size_t m = (lo + hi) / 2; if(KO[m].mer < mer) lo = m + 1; else hi = m; } size_t e = lo; while(e < kn && KO[e].mer == mer) e++;
			size_t c = e - lo; if(c > 0xFFFF) c = 0xFFFF; if(c > K || c <= 1) continue; nkh++; if(c > maxrun) maxrun = c; uint32_t ln = qend - qoff; Tall += c;
			for(size_t x = lo; x < lo + c; x++){ uint32_t sd = KO[x].sd; if((sd >> 1) == r->id) continue; if(R[sd >> 1].len > up) continue; if(tn == tc){ tc *= 2; T = realloc(T, tc * sizeof(tup_t)); } T[tn].sd = sd; T[tn].qoff = qoff; T[tn].len = ln; tn++; } });
		qsort(T, tn, sizeof(tup_t), cmp_tup);
		uint32_t G = 0, Gs = 0, Go = 0;
		for(size_t a = 0; a < tn; ){ size_t b = a; uint32_t ol = 0, lst = 0; uint64_t sum = 0; while(b < tn && T[b].sd == T[a].sd){ sum += T[b].len; if(T[b].qoff >= lst) ol += T[b].len; else ol += T[b].qoff + T[b].len - lst; lst = T[b].qoff + T[b].len; b++; } G++; if(sum >= kovl) Gs++; if(ol >= kovl) Go++; a = b; }
		printf("%u\t%u\t%u\t%u\t%zu\t%zu\t%u\t%u\t%u\t%u\n", Q[k], r->len, nk, nkh, Tall, tn, G, Gs, Go, maxrun);
	}
	return 0;
}
