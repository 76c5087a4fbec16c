/*
 * wtz_host.h — host-side (plain C) pieces of the drop-in `wtzmo` executable: read ingest, the
 * reference's exact unstable sort for the three host-side orderings whose ties are observable
 * (read ids wtzmo.c:1708, candidate order 821, seed order 986), small containers.
 *
 * The device work is reached only through include/wtzmo_hip.h (libwtzmo_hip.so).  There is no CPU
 * implementation of the hot path in this program: without a GPU it exits with an error.
 */
#ifndef WTZ_HOST_H
#define WTZ_HOST_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "wtzmo_hip.h"

static inline void *hx_realloc(void *p, size_t n){
	void *q = realloc(p, n ? n : 1);
	if(!q){ fprintf(stderr, " -- Out of memory, try to allocate %zu bytes --\n", n); exit(1); }
	return q;
}

/* growable byte buffer */
typedef struct { char *s; size_t n, cap; } hx_str_t;
static inline void hx_str_add(hx_str_t *b, const char *p, size_t k){
	if(b->n + k + 1 > b->cap){ size_t c = b->cap ? b->cap : 256; while(c < b->n + k + 1) c <<= 1; b->s = (char*)hx_realloc(b->s, c); b->cap = c; }
	memcpy(b->s + b->n, p, k); b->n += k; b->s[b->n] = 0;
}

/*
 * The reference's sort (sort.h:104-155) as a generic routine over fixed-size records.
 * Observable tie order => the swap sequence is part of the contract: median-of-three by up to three
 * swaps, Hoare scan against a pivot copy, ranges under six elements left for a tail-to-head bubble
 * pass, larger side pushed first.  gt(a,b,ctx) != 0 iff a is "greater than" b at the call site.
 */
typedef int (*hx_gt_fn)(const void *a, const void *b, void *ctx);
static void hx_sort_exact(void *base, size_t n, size_t sz, hx_gt_fn gt, void *ctx){
	if(n < 2) return;
	unsigned char *v = (unsigned char*)base;
	unsigned char tmp[64], piv[64];
	if(sz > sizeof tmp){ fprintf(stderr, "hx_sort_exact: record too large\n"); exit(1); }
#define HX_AT(i) (v + (i) * sz)
#define HX_SWAP(i, j) do { memcpy(tmp, HX_AT(i), sz); memcpy(HX_AT(i), HX_AT(j), sz); memcpy(HX_AT(j), tmp, sz); } while(0)
	size_t lo[64], hi[64]; int sp = 0;
	lo[0] = 0; hi[0] = n - 1; sp = 1;
	while(sp > 0){
		sp--;
		const size_t s = lo[sp], e = hi[sp], m = s + (e - s) / 2;
		if(gt(HX_AT(s), HX_AT(m), ctx)) HX_SWAP(s, m);
		if(gt(HX_AT(m), HX_AT(e), ctx)){ HX_SWAP(e, m); if(gt(HX_AT(s), HX_AT(m), ctx)) HX_SWAP(s, m); }
		memcpy(piv, HX_AT(m), sz);
		size_t i = s + 1, j = e - 1;
		while(1){
			while(gt(piv, HX_AT(i), ctx)) i++;
			while(gt(HX_AT(j), piv, ctx)) j--;
			if(i < j){ HX_SWAP(i, j); i++; j--; } else break;
		}
		if(i == j){ i++; j--; }
		const int left_big = (j - s > e - i);
		const int push_l = (s + 4 < j), push_r = (i + 4 < e);
		if(left_big){ if(push_l){ lo[sp] = s; hi[sp] = j; sp++; } if(push_r){ lo[sp] = i; hi[sp] = e; sp++; } }
		else        { if(push_r){ lo[sp] = i; hi[sp] = e; sp++; } if(push_l){ lo[sp] = s; hi[sp] = j; sp++; } }
	}
	for(size_t i = 0; i < n; i++){
		int moved = 0;
		for(size_t j = n - 1; j > i; j--) if(gt(HX_AT(j - 1), HX_AT(j), ctx)){ HX_SWAP(j - 1, j); moved = 1; }
		if(!moved) break;
	}
#undef HX_AT
#undef HX_SWAP
}

/* the same routine for one record type with the comparison inlined (same swap sequence: the order of equal keys is part of the output).  GT(a, b) takes two
 * `const T*`.  The generic form above costs a call through a pointer and three memcpy per swap: 0.05 s of the sequential commit per configs[2] step went there. */
#define HX_DEFINE_SORT_EXACT(FNAME, T, GT) \
static void FNAME(T *v, size_t n){ \
	if(n < 2) return; \
	size_t lo[64], hi[64]; int sp = 0; T tmp, piv; \
	lo[0] = 0; hi[0] = n - 1; sp = 1; \
	while(sp > 0){ \
		sp--; \
		const size_t s = lo[sp], e = hi[sp], m = s + (e - s) / 2; \
		if(GT(&v[s], &v[m])){ tmp = v[s]; v[s] = v[m]; v[m] = tmp; } \
		if(GT(&v[m], &v[e])){ tmp = v[e]; v[e] = v[m]; v[m] = tmp; if(GT(&v[s], &v[m])){ tmp = v[s]; v[s] = v[m]; v[m] = tmp; } } \
		piv = v[m]; \
		size_t i = s + 1, j = e - 1; \
		while(1){ \
			while(GT(&piv, &v[i])) i++; \
			while(GT(&v[j], &piv)) j--; \
			if(i < j){ tmp = v[i]; v[i] = v[j]; v[j] = tmp; i++; j--; } else break; \
		} \
		if(i == j){ i++; j--; } \
		const int left_big = (j - s > e - i); \
		const int push_l = (s + 4 < j), push_r = (i + 4 < e); \
		if(left_big){ if(push_l){ lo[sp] = s; hi[sp] = j; sp++; } if(push_r){ lo[sp] = i; hi[sp] = e; sp++; } } \
		else        { if(push_r){ lo[sp] = i; hi[sp] = e; sp++; } if(push_l){ lo[sp] = s; hi[sp] = j; sp++; } } \
	} \
	for(size_t i = 0; i < n; i++){ \
		int moved = 0; \
		for(size_t j = n - 1; j > i; j--) if(GT(&v[j - 1], &v[j])){ tmp = v[j - 1]; v[j - 1] = v[j]; v[j] = tmp; moved = 1; } \
		if(!moved) break; \
	} \
}

/* ---------------- reads ---------------- */
typedef struct { uint64_t off; uint32_t len; char *name; } hx_read_t;     /* pbread_t, wtzmo.c:87-90 */

typedef struct {
	uint64_t *bits; uint64_t nbase, capw;      /* BaseBank layout, dna.h:318-322 */
	hx_read_t *reads; uint32_t n_all, cap_reads;
	uint32_t n_rd, n_qr;                        /* indexed reads / -I query-only reads */
	int keep_text; char *text; uint64_t captext; /* f4: keep the bases as TEXT (file order) for wtz_upload_reads_ascii instead of packing them here */
} hx_store_t;

static inline void hx_store_put(hx_store_t *st, unsigned b){
	const uint64_t i = st->nbase;
	if((i >> 5) >= st->capw){
		uint64_t c = st->capw ? st->capw * 2 : 4096;
		st->bits = (uint64_t*)hx_realloc(st->bits, c * 8);
		memset(st->bits + st->capw, 0, (c - st->capw) * 8);
		st->capw = c;
	}
	st->bits[i >> 5] |= ((uint64_t)(b & 3u)) << (((~i) & 31u) << 1);
	st->nbase = i + 1;
}

static void hx_store_add(hx_store_t *st, const char *name, size_t nlen, const char *seq, size_t slen){
	static const signed char code[256] = {
		['A'] = 1, ['a'] = 1, ['C'] = 2, ['c'] = 2, ['G'] = 3, ['g'] = 3, ['T'] = 4, ['t'] = 4 };
	if(st->n_all == st->cap_reads){ st->cap_reads = st->cap_reads ? st->cap_reads * 2 : 1024; st->reads = (hx_read_t*)hx_realloc(st->reads, sizeof(hx_read_t) * st->cap_reads); }
	hx_read_t *r = &st->reads[st->n_all++];
	r->off = st->nbase; r->len = (uint32_t)slen;
	r->name = (char*)hx_realloc(NULL, nlen + 1); memcpy(r->name, name, nlen); r->name[nlen] = 0;
	if(st->keep_text){       /* packed (and its non-ACGT bytes drawn) on the device */
		if(st->nbase + slen + 1 > st->captext){ uint64_t c = st->captext ? st->captext : ((uint64_t)1 << 20); while(c < st->nbase + slen + 1) c += c / 2; st->text = (char*)hx_realloc(st->text, c); st->captext = c; }
		memcpy(st->text + st->nbase, seq, slen); st->nbase += slen;
		return;
	}
	for(size_t i = 0; i < slen; i++){
		int c = code[(unsigned char)seq[i]];
		unsigned b = c ? (unsigned)(c - 1) : (unsigned)(lrand48() & 3);      /* non-ACGT: dna.h:405, file order, default seed */
		hx_store_put(st, b);
	}
}

/* ---------------- sequence file reader (file_reader.c:296-416, 66-71) ---------------- */
typedef struct {
	char **paths; int npath, cur; FILE *fp; int is_proc;
	char *line; size_t cap; long n; int pushed; int kind;
} hx_reader_t;

static int hx_reader_next_file(hx_reader_t *r){
	if(r->cur >= r->npath) return 0;
	const char *p = r->paths[r->cur]; const size_t L = strlen(p);
	if(strcmp(p, "-") == 0){ r->fp = stdin; r->is_proc = 0; return 1; }
	if(L > 3 && strcmp(p + L - 3, ".gz") == 0){
		char *cmd = (char*)hx_realloc(NULL, L + 32); sprintf(cmd, "gzip -dc %s", p);
		r->fp = popen(cmd, "r"); r->is_proc = 1; free(cmd);
	} else { r->fp = fopen(p, "r"); r->is_proc = 0; }
	return r->fp != NULL;
}
static hx_reader_t *hx_reader_open(char **paths, int npath){
	hx_reader_t *r = (hx_reader_t*)calloc(1, sizeof *r);
	r->paths = paths; r->npath = npath;
	if(!hx_reader_next_file(r)){ free(r); return NULL; }
	return r;
}
static void hx_reader_close(hx_reader_t *r){
	if(r->fp && r->fp != stdin){ if(r->is_proc) pclose(r->fp); else fclose(r->fp); }
	free(r->line); free(r);
}
static long hx_reader_line(hx_reader_t *r){
	if(r->pushed){ r->pushed = 0; return r->n; }
	size_t n = 0; int got = 0;
	while(r->fp){
		int c = getc_unlocked(r->fp);
		if(c == EOF){
			if(r->fp != stdin){ if(r->is_proc) pclose(r->fp); else fclose(r->fp); }
			r->fp = NULL; r->cur++;
			if(r->cur < r->npath && hx_reader_next_file(r)) continue;
			break;
		}
		got = 1;
		if(c == '\n') break;
		if(n + 2 > r->cap){ r->cap = r->cap ? r->cap * 2 : 4096; r->line = (char*)hx_realloc(r->line, r->cap); }
		r->line[n++] = (char)c;
	}
	if(!got){ r->n = -1; return -1; }
	if(n + 1 > r->cap){ r->cap = n + 1; r->line = (char*)hx_realloc(r->line, r->cap); }
	r->line[n] = 0; r->n = (long)n;
	return r->n;
}
/* returns 1 and fills name/seq for the next record, 0 at end; desc (may be NULL) = the rest of the header line behind the name
 * (Sequence.header + tag.size, file_reader.c:327-328, 371-372) */
static int hx_reader_seq_desc(hx_reader_t *r, hx_str_t *name, hx_str_t *desc, hx_str_t *seq){
	long n;
	if(r->kind == 0){
		r->kind = 3;
		while((n = hx_reader_line(r)) != -1){
			if(n == 0 || r->line[0] == '#') continue;
			r->kind = r->line[0] == '>' ? 1 : (r->line[0] == '@' ? 2 : 3);
			r->pushed = 1; break;
		}
	}
	name->n = 0; seq->n = 0; if(desc) desc->n = 0;
	if(r->kind == 1){
		int state = 0;
		while((n = hx_reader_line(r)) != -1){
			if(n && r->line[0] == '>'){
				if(state){ r->pushed = 1; break; }
				state = 1;
				long i = 1; while(i < n && r->line[i] != ' ' && r->line[i] != '\t' && r->line[i] != '\r' && r->line[i] != '\n') i++;
				hx_str_add(name, r->line + 1, (size_t)(i - 1));
				if(desc) hx_str_add(desc, r->line + i, (size_t)(n - i));
			} else if(state){ hx_str_add(seq, r->line, (size_t)n); state = 2; }
		}
		return state != 0;
	}
	if(r->kind == 2){
		int state = 0;
		while(state != 4 && (n = hx_reader_line(r)) >= 0){
			if(state == 0){ if(r->line[0] != '@') continue; state = 1;
				long i = 1; while(i < n && r->line[i] != ' ' && r->line[i] != '\t' && r->line[i] != '\n') i++;
				hx_str_add(name, r->line + 1, (size_t)(i - 1)); if(desc) hx_str_add(desc, r->line + i, (size_t)(n - i)); }
			else if(state == 1){ state = 2; hx_str_add(seq, r->line, (size_t)n); }
			else if(state == 2){ if(r->line[0] == '+') state = 3; }
			else state = 4;
		}
		return state == 4;
	}
	return 0;
}

static int hx_reader_seq(hx_reader_t *r, hx_str_t *name, hx_str_t *seq){ return hx_reader_seq_desc(r, name, NULL, seq); }

/* ---------------- open-addressing u64 set / name map ---------------- */
typedef struct { uint64_t *tab; size_t cap, n; } hx_set_t;
static inline uint64_t hx_mix(uint64_t x){ x ^= x >> 31; x *= 0x7fb5d329728ea185ULL; x ^= x >> 27; x *= 0x81dadef4bc2dd44dULL; x ^= x >> 33; return x; }
static int hx_set_has(const hx_set_t *s, uint64_t v){
	if(!s->cap) return 0;
	size_t m = s->cap - 1, i = hx_mix(v) & m;
	while(s->tab[i] != ~0ULL){ if(s->tab[i] == v) return 1; i = (i + 1) & m; }
	return 0;
}
/* the same probe for a thread that reads while the owner inserts (round 6: the commit's helper threads).  Safe as long as the table does not grow meanwhile
 * (hx_set_reserve in front): a slot goes from empty to a key in one aligned 64-bit store, so a reader sees the key or not yet - callers detect "not yet" by
 * other means (wtzmo_main.c: closed_touch) */
static int hx_set_has_racy(const hx_set_t *s, uint64_t v){
	if(!s->cap) return 0;
	size_t m = s->cap - 1, i = hx_mix(v) & m;
	for(;;){ const uint64_t x = __atomic_load_n(&s->tab[i], __ATOMIC_RELAXED); if(x == ~0ULL) return 0; if(x == v) return 1; i = (i + 1) & m; }
}
static int hx_set_put(hx_set_t *s, uint64_t v);
/* room for `extra` more keys without a growth step */
static void hx_set_reserve(hx_set_t *s, size_t extra){
	while((s->n + extra + 1) * 2 > s->cap){
		size_t oc = s->cap; uint64_t *ot = s->tab;
		s->cap = oc ? oc * 2 : 4096; s->n = 0;
		s->tab = (uint64_t*)hx_realloc(NULL, s->cap * 8); memset(s->tab, 0xFF, s->cap * 8);
		for(size_t k = 0; k < oc; k++) if(ot[k] != ~0ULL) hx_set_put(s, ot[k]);
		free(ot);
	}
}
static int hx_set_put(hx_set_t *s, uint64_t v){       /* returns 1 if newly inserted */
	if((s->n + 1) * 2 > s->cap){
		size_t oc = s->cap; uint64_t *ot = s->tab;
		s->cap = oc ? oc * 2 : 4096; s->n = 0;
		s->tab = (uint64_t*)hx_realloc(NULL, s->cap * 8); memset(s->tab, 0xFF, s->cap * 8);
		for(size_t k = 0; k < oc; k++) if(ot[k] != ~0ULL) hx_set_put(s, ot[k]);
		free(ot);
	}
	size_t m = s->cap - 1, i = hx_mix(v) & m;
	while(s->tab[i] != ~0ULL){ if(s->tab[i] == v) return 0; i = (i + 1) & m; }
	__atomic_store_n(&s->tab[i], v, __ATOMIC_RELAXED); s->n++; return 1;
}
/* ---- the ORDER of the -9 pair file.  The reference writes closed_alns in the iteration order of its own hash set (wtzmo.c:1797-1801):
 * slot order of an open-addressing table (hashset.h:64-432: linear probing, Jenkins 64-bit hash of the key modulo a size taken from a
 * fixed list, growth at load 0.67 by an in-place re-insertion that displaces not-yet-moved keys).  The slot a key ends up in depends only
 * on the sequence of insertions, so the file is reproduced by replaying that sequence into a model of the table that tracks slots only. */
static const uint64_t hx_ref_sizes[] = {      /* hashset.h:30-44 (sys_prime_list): the table sizes the reference can take - a format constant */
	0x7ULL, 0xfULL, 0x1fULL, 0x43ULL, 0x89ULL, 0x115ULL, 0x22dULL, 0x45dULL, 0x8bdULL, 0x1181ULL, 0x2303ULL, 0x4609ULL, 0x8c17ULL, 0x1183dULL, 0x2307bULL,
	0x460fdULL, 0x8c201ULL, 0x118411ULL, 0x230833ULL, 0x461069ULL, 0x8c20e1ULL, 0x11841cbULL, 0x2308397ULL, 0x461075bULL, 0x8c20ecbULL, 0x11841da5ULL,
	0x23083b61ULL, 0x461076c7ULL, 0x8c20ed91ULL, 0x11841db31ULL, 0x23083b673ULL, 0x461076d1bULL, 0x8c20eda41ULL, 0x11841db48dULL, 0x23083b6937ULL,
	0x461076d27fULL, 0x8c20eda50dULL, 0x11841db4a59ULL, 0x23083b694ebULL, 0x461076d29f1ULL, 0x8c20eda5441ULL, 0x11841db4a887ULL, 0x23083b69511fULL,
	0x461076d2a2c1ULL, 0x8c20eda54591ULL, 0x11841db4a8b55ULL, 0x23083b69516c1ULL, 0x461076d2a2da5ULL, 0x8c20eda545b55ULL, 0x11841db4a8b6b5ULL,
	0x23083b69516d91ULL, 0x461076d2a2db3bULL, 0x8c20eda545b69dULL, 0x11841db4a8b6d5dULL, 0x23083b69516daf5ULL, 0x461076d2a2db5edULL,
	0x8c20eda545b6c5fULL, 0x11841db4a8b6d8ebULL, 0x23083b69516db1ffULL, 0x461076d2a2db643fULL, 0x8c20eda545b6c8f3ULL };
static uint64_t hx_ref_size_for(uint64_t n){
	const size_t last = sizeof hx_ref_sizes / sizeof hx_ref_sizes[0] - 1; size_t i = 0;
	while(i < last && n > hx_ref_sizes[i]) i++;
	return hx_ref_sizes[i];
}
static inline uint64_t hx_ref_hash(uint64_t k){       /* hashset.h:464-474 */
	k += ~(k << 32); k ^= (k >> 22); k += ~(k << 13); k ^= (k >> 8); k += (k << 3); k ^= (k >> 15); k += ~(k << 27); k ^= (k >> 31);
	return k;
}
typedef struct { uint64_t *slot; uint8_t *full; uint64_t size, count, limit; } hx_refslots_t;
static void hx_refslots_init(hx_refslots_t *t, uint64_t hint){
	t->size = hx_ref_size_for(hint); t->count = 0; t->limit = (uint64_t)((float)t->size * 0.67f);
	t->slot = (uint64_t*)hx_realloc(NULL, 8 * t->size); t->full = (uint8_t*)calloc(t->size, 1);
}
static void hx_refslots_grow(hx_refslots_t *t){
	uint64_t n = t->size;
	do { n = hx_ref_size_for(n * 2); } while((float)n * 0.67f < (float)(t->count + 1));
	const uint64_t old = t->size;
	t->slot = (uint64_t*)hx_realloc(t->slot, 8 * n);
	uint8_t *waiting = t->full;                      /* keys still sitting at their OLD slot */
	uint8_t *full = (uint8_t*)calloc(n, 1);
	for(uint64_t i = 0; i < old; i++){
		if(!waiting[i]) continue;
		uint64_t key = t->slot[i]; waiting[i] = 0;
		for(;;){
			uint64_t h = hx_ref_hash(key) % n;
			while(full[h]) h = (h + 1) % n;
			full[h] = 1;
			if(h < old && waiting[h]){ const uint64_t evicted = t->slot[h]; t->slot[h] = key; key = evicted; waiting[h] = 0; }      /* its turn comes now */
			else { t->slot[h] = key; break; }
		}
	}
	free(waiting);
	t->full = full; t->size = n; t->limit = (uint64_t)((float)n * 0.67f);
}
/* a put of a key that is already there: the reference's put_u64hash runs its capacity check BEFORE looking the key up (hashset.h:224-226, 351),
 * so a duplicate arriving at a full table grows it one step earlier than the next new key would */
static void hx_refslots_touch(hx_refslots_t *t){ if(t->count + 1 > t->limit) hx_refslots_grow(t); }
static void hx_refslots_put(hx_refslots_t *t, uint64_t key){      /* new keys only; duplicates go through hx_refslots_touch */
	if(t->count + 1 > t->limit) hx_refslots_grow(t);
	uint64_t h = hx_ref_hash(key) % t->size;
	while(t->full[h]) h = h + 1 == t->size ? 0 : h + 1;
	t->full[h] = 1; t->slot[h] = key; t->count++;
}

static inline uint64_t hx_pair_key(uint64_t a, uint64_t b){ return a < b ? ((a << 33) | (b << 1)) : ((b << 33) | (a << 1)); }   /* wtzmo.c:83-84 */

typedef struct { uint32_t *tab; size_t cap; const hx_read_t *reads; } hx_names_t;
static uint64_t hx_strhash(const char *s){ uint64_t h = 1469598103934665603ULL; while(*s){ h ^= (unsigned char)*s++; h *= 1099511628211ULL; } return h; }
static void hx_names_build(hx_names_t *m, const hx_read_t *reads, uint32_t n){
	m->cap = 16; while(m->cap < (size_t)n * 2 + 2) m->cap <<= 1;
	m->tab = (uint32_t*)hx_realloc(NULL, m->cap * 4); memset(m->tab, 0xFF, m->cap * 4); m->reads = reads;
	for(uint32_t i = 0; i < n; i++){
		size_t k = hx_strhash(reads[i].name) & (m->cap - 1);
		while(m->tab[k] != 0xFFFFFFFFu && strcmp(reads[m->tab[k]].name, reads[i].name)) k = (k + 1) & (m->cap - 1);
		m->tab[k] = i;
	}
}
static uint32_t hx_names_get(const hx_names_t *m, const char *s){
	size_t k = hx_strhash(s) & (m->cap - 1);
	while(m->tab[k] != 0xFFFFFFFFu){ if(strcmp(m->reads[m->tab[k]].name, s) == 0) return m->tab[k]; k = (k + 1) & (m->cap - 1); }
	return 0xFFFFFFFFu;
}

#endif
