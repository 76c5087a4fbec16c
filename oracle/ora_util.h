/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by or executed from the
 * product path (smartdenovo_amd/, include/).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use anything under oracle/.
 *
 * ora_util.h — growable vectors plus the two container behaviours of the reference whose
 * *tie order* is observable in wtzmo's output and therefore has to be restated exactly:
 *
 *   - the unstable quicksort + bubble pass          (reference sort.h:104-155, `sort_array`)
 *   - the binary heap sift-up / sift-down rules      (reference list.h:78-144, `array_heap_*`)
 *
 * Parity of this restatement is pinned against the real reference (oracle/_ref, built by
 * oracle/Makefile from /root/reference) by tests/test_oracle_vs_reference.py and against the
 * committed goldens under tests/golden/.
 */
#ifndef ORA_UTIL_H
#define ORA_UTIL_H

#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

static inline void *ora_xrealloc(void *p, size_t n){
	void *q = realloc(p, n ? n : 1);
	if(q == NULL){ fprintf(stderr, "oracle: out of memory (%zu bytes)\n", n); exit(1); }
	return q;
}

/* A minimal typed vector: ORA_VEC(name, T) defines `name` {T *a; size_t n, cap;} */
#define ORA_VEC(NAME, T) \
typedef struct { T *a; size_t n, cap; } NAME; \
static inline void NAME##_reserve(NAME *v, size_t want){ \
	if(want > v->cap){ size_t c = v->cap ? v->cap : 16; while(c < want) c <<= 1; \
		v->a = (T*)ora_xrealloc(v->a, c * sizeof(T)); v->cap = c; } } \
static inline void NAME##_push(NAME *v, T x){ NAME##_reserve(v, v->n + 1); v->a[v->n++] = x; } \
static inline T *NAME##_next(NAME *v){ NAME##_reserve(v, v->n + 1); return &v->a[v->n++]; } \
static inline void NAME##_append(NAME *v, const T *src, size_t k){ \
	if(k){ NAME##_reserve(v, v->n + k); memcpy(v->a + v->n, src, k * sizeof(T)); v->n += k; } } \
static inline void NAME##_free(NAME *v){ free(v->a); v->a = NULL; v->n = v->cap = 0; }

ORA_VEC(vec_u8,  uint8_t)
ORA_VEC(vec_u16, uint16_t)
ORA_VEC(vec_u32, uint32_t)
ORA_VEC(vec_i32, int32_t)
ORA_VEC(vec_u64, uint64_t)
ORA_VEC(vec_f32, float)

/*
 * Exact restatement of the reference's sort (sort.h:104-155).  The sort is NOT stable and
 * its tie order leaks into wtzmo's output (SURVEY §8a "exactness traps" 1), so the precise
 * sequence of swaps is part of the contract:
 *   1. iterative quicksort with an explicit range stack; median-of-three arranged by up to
 *      three swaps among {lo, mid, hi}; Hoare scan from lo+1 / hi-1 against a *copy* of the
 *      pivot; after the scan `i==j` steps both; sub-ranges spanning fewer than six elements
 *      are left unsorted; the larger side is pushed first (so the smaller is popped first);
 *   2. a tail-to-head bubble pass repeated until a pass makes no swap.
 * GT(a, b) must evaluate to >0 iff "a is greater than b" exactly as the reference's
 * comparison expression at that call site does.
 */
#define ORA_DEFINE_SORT(NAME, T, GT) \
static void NAME(T *v, size_t n, void *ctx){ \
	(void)ctx; \
	if(n < 2) return; \
	size_t lo_stk[64], hi_stk[64]; int sp = 0; \
	T piv, tmp; \
	lo_stk[sp] = 0; hi_stk[sp] = n - 1; sp++; \
	while(sp){ \
		sp--; \
		size_t s = lo_stk[sp], e = hi_stk[sp], m = s + (e - s) / 2; \
		if((GT(v[s], v[m])) > 0){ tmp = v[s]; v[s] = v[m]; v[m] = tmp; } \
		if((GT(v[m], v[e])) > 0){ \
			tmp = v[e]; v[e] = v[m]; v[m] = tmp; \
			if((GT(v[s], v[m])) > 0){ tmp = v[s]; v[s] = v[m]; v[m] = tmp; } \
		} \
		piv = v[m]; \
		size_t i = s + 1, j = e - 1; \
		for(;;){ \
			while((GT(piv, v[i])) > 0) i++; \
			while((GT(v[j], piv)) > 0) j--; \
			if(i < j){ tmp = v[i]; v[i] = v[j]; v[j] = tmp; i++; j--; } \
			else break; \
		} \
		if(i == j){ i++; j--; } \
		if(j - s > e - i){ \
			if(s + 4 < j){ lo_stk[sp] = s; hi_stk[sp] = j; sp++; } \
			if(i + 4 < e){ lo_stk[sp] = i; hi_stk[sp] = e; sp++; } \
		} else { \
			if(i + 4 < e){ lo_stk[sp] = i; hi_stk[sp] = e; sp++; } \
			if(s + 4 < j){ lo_stk[sp] = s; hi_stk[sp] = j; sp++; } \
		} \
	} \
	for(size_t i = 0; i < n; i++){ \
		int swapped = 0; \
		for(size_t j = n - 1; j > i; j--){ \
			if((GT(v[j-1], v[j])) > 0){ tmp = v[j-1]; v[j-1] = v[j]; v[j] = tmp; swapped = 1; } \
		} \
		if(!swapped) break; \
	} \
}

/*
 * Exact restatement of the reference heap (list.h:78-144): a min-heap under CMP where
 * sift-up stops at the first parent with CMP(child,parent) >= 0 and sift-down prefers the
 * left child, taking the right one only if it is strictly smaller than the current choice.
 * CMP(a,b) returns <0, 0, >0.
 */
#define ORA_DEFINE_HEAP(NAME, T, CMP) \
static inline void NAME##_push(T *h, size_t *n, T x, void *ctx){ \
	(void)ctx; size_t i = (*n)++; h[i] = x; \
	while(i){ size_t p = (i - 1) >> 1; if((CMP(h[i], h[p])) >= 0) break; \
		T t = h[i]; h[i] = h[p]; h[p] = t; i = p; } } \
static inline void NAME##_sift(T *h, size_t n, size_t idx, void *ctx){ \
	(void)ctx; \
	while((idx << 1) + 1 < n){ size_t pick = idx, l = (idx << 1) + 1, r = l + 1; \
		if((CMP(h[pick], h[l])) > 0) pick = l; \
		if(r < n && (CMP(h[pick], h[r])) > 0) pick = r; \
		if(pick == idx) break; \
		T t = h[idx]; h[idx] = h[pick]; h[pick] = t; idx = pick; } } \
static inline void NAME##_replace_top(T *h, size_t n, T x, void *ctx){ h[0] = x; NAME##_sift(h, n, 0, ctx); } \
static inline void NAME##_remove_top(T *h, size_t *n, void *ctx){ h[0] = h[--(*n)]; NAME##_sift(h, *n, 0, ctx); }

#define ORA_MIN(a,b) ((a) < (b) ? (a) : (b))
#define ORA_MAX(a,b) ((a) > (b) ? (a) : (b))
#define ORA_ABSDIFF(a,b) ((a) < (b) ? (b) - (a) : (a) - (b))

#endif
