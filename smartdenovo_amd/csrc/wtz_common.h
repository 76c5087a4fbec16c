/*
 * wtz_common.h — shared device/host definitions of the MI355X wtzmo hot path.
 *
 * Everything in csrc/wtz_*.h is written as plain C++ functions marked WTZ_HD so that
 *   (a) hipcc compiles them into the gfx950 kernels of libwtzmo_hip.so (the product), and
 *   (b) tests/emul/ can compile the very same task bodies with g++ and run each "kernel" as
 *       a host loop, which lets the host driver, batching and commit logic be debugged in a
 *       container without a GPU.  (b) is test infrastructure only; nothing in the product
 *       loads it and the product fails loudly when the HIP library is missing.
 */
#ifndef WTZ_COMMON_H
#define WTZ_COMMON_H

#include <stdint.h>
#include <stddef.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define WTZ_HD __host__ __device__ __forceinline__
#define WTZ_HDM __host__ __device__ __forceinline__
#define WTZ_HDN __host__ __device__ __noinline__
#define WTZ_DN  __device__ __noinline__
#define WTZ_D  __device__ __forceinline__
#else
#define WTZ_HD static inline
#define WTZ_HDM inline
#define WTZ_HDN static
#define WTZ_DN  static
#define WTZ_D  static inline
#endif

#define WTZ_MIN(a,b) ((a) < (b) ? (a) : (b))
#define WTZ_MAX(a,b) ((a) > (b) ? (a) : (b))
#define WTZ_ABSDIFF(a,b) ((a) < (b) ? (b) - (a) : (a) - (b))

/* ---------------- parameters: the public ABI struct (include/wtzmo_hip.h) ---------------- */
#include "wtzmo_hip.h"
typedef wtz_params_c wtz_params_t;

/* ---------------- record types ---------------- */
/* z-mer match, 16 B like the reference's hzmp_t (hzm_aln.h:54-59): o1 = dir1<<31 | off1, o2 = dir2<<31 | off2,
 * ll = len2<<16 | len1, gid = group id (dot-matrix engine) */
typedef struct { uint32_t o1, o2, ll, gid; } wtz_zhit_t;
#define ZH_OFF1(h) ((h).o1 & 0x7FFFFFFFu)
#define ZH_OFF2(h) ((h).o2 & 0x7FFFFFFFu)
#define ZH_DIR1(h) ((h).o1 >> 31)
#define ZH_DIR2(h) ((h).o2 >> 31)
#define ZH_LEN1(h) ((h).ll & 0xFFFFu)
#define ZH_LEN2(h) ((h).ll >> 16)
#define ZH_STRAND(h) (((h).o1 ^ (h).o2) >> 31)       /* dir1 ^ dir2 */

typedef struct {
	uint32_t pb2;
	uint32_t ovl;          /* :29 in the reference */
	uint8_t  dir, closed; uint16_t pad;
	int32_t  beg[2], end[2];
	uint32_t anchors[2];
} wtz_win_t;               /* wt_seed_t (hzm_aln.h:62-67) */

typedef struct { int32_t score, tb, te, qb, qe, aln, mat, mis, ins, del; } wtz_aln_t;   /* kswx_t (kswx.h:30-34) */

#define WTZ_OVL29(x) ((uint32_t)(x) & 0x1FFFFFFFu)

/* ---------------- 2-bit reads (dna.h:78: base i at bits ((~i)&31)*2 of word i>>5) ---------------- */
WTZ_HD uint32_t wtz_base_at(const uint64_t *bits, uint64_t i){
	return (uint32_t)((bits[i >> 5] >> (((~i) & 31u) << 1)) & 3u);
}

/* reverse complement of a k-mer held in the low 2k bits (dna.h:85-98) */
WTZ_HD uint64_t wtz_revcomp_kmer(uint64_t x, unsigned k){
	x = ~x;
	x = ((x & 0x3333333333333333ULL) << 2) | ((x & 0xCCCCCCCCCCCCCCCCULL) >> 2);
	x = ((x & 0x0F0F0F0F0F0F0F0FULL) << 4) | ((x & 0xF0F0F0F0F0F0F0F0ULL) >> 4);
	x = ((x & 0x00FF00FF00FF00FFULL) << 8) | ((x & 0xFF00FF00FF00FF00ULL) >> 8);
	x = ((x & 0x0000FFFF0000FFFFULL) << 16) | ((x & 0xFFFF0000FFFF0000ULL) >> 16);
	x = (x << 32) | (x >> 32);
	return x >> (64 - (k << 1));
}

WTZ_HD uint32_t wtz_jenkins32(uint32_t key){      /* hashset.h:452-462, k-mer subsample hash (wtzmo.c:35) */
	key += (key << 12); key ^= (key >> 22);
	key += (key << 4);  key ^= (key >> 9);
	key += (key << 10); key ^= (key >> 2);
	key += (key << 7);  key ^= (key >> 12);
	return key;
}

WTZ_HD uint64_t wtz_mix64(uint64_t x){ x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

/* ---------------- device bump pool ----------------
 * One large HBM arena per context; tasks carve scratch and result arrays out of it.  Nothing is freed inside a stage; the host resets
 * the pool between stages.  Exhaustion sets `overflow` (the stage then reports WTZ_E_POOL, loudly).
 *
 * Round 5: the counter is no longer ONE word.  Through round 4 every allocation was a device-scope atomic add on `used`; a configs[2] step
 * makes ~10^7 of them from ~2 000 resident waves, and same-address atomics serialise at a few tens of nanoseconds each: a K-sw3 job that
 * executes ONE row took as long as 100 rows (measured: 40 000 one-row jobs 8.5 ms with the single counter, 0.16 ms without), the dmo pair
 * kernel lost a fifth of its time there.  Now the global cursor `used` hands out SLABS (64 KB ... 2 MB by pool size) to WTZ_POOL_NSHARD
 * shard cursors, each in its own 128-byte line and picked by the workgroup index; an allocation is one atomic add on its shard:
 *   state = slab tag (slab offset / 256 + 1, 0 = none) << 32 | bytes taken of the slab.
 * The add that CROSSES the slab's end (taken <= slab < taken + n: exactly one request per slab) fetches the next slab from the global
 * cursor, serves itself from its start and publishes it with an exchange; requests that arrive on an exhausted slab before that are served
 * from the global cursor directly, as are requests larger than a quarter slab.  No locks, no spinning (lanes of one wave allocate side by side).
 * `used` = bytes handed out including slabs in progress: an upper bound of what is in use, at most NSHARD slabs above it. */
#define WTZ_POOL_NSHARD 64
typedef struct {
	uint8_t *base;
	unsigned long long cap;
	unsigned long long used;
	int overflow;
	/* fault injection (WTZ_POOL_FAIL_AT=<n>, tests only): the n-th and every later request of a stage fails as if the pool were full */
	unsigned int fail_at, nalloc;
	unsigned long long slab;                                   /* bytes per slab (a multiple of 256); 0 = every request goes to the global cursor */
	unsigned long long shard[WTZ_POOL_NSHARD * 16];             /* one cursor per 128-byte line */
} wtz_pool_t;

/* host side: a pool over [base, base + cap) with nothing handed out */
static inline void wtz_pool_init(wtz_pool_t *p, uint8_t *base, unsigned long long cap, unsigned int fail_at){
	memset(p, 0, sizeof *p);
	p->base = base; p->cap = cap; p->fail_at = fail_at;
	unsigned long long s = (unsigned long long)64 << 10;
	while(s < ((unsigned long long)2 << 20) && s * 2048 <= cap) s <<= 1;
	p->slab = cap >= ((unsigned long long)128 << 20) ? s : 0;      /* 64 shards of >= 64 KB each are up to 4 MB of slack on top of a planned budget: small pools keep the single cursor */
	for(int k = 0; k < WTZ_POOL_NSHARD; k++) p->shard[k * 16] = p->slab;      /* tag 0, slab "taken" to its end: the first request crosses */
}

WTZ_HD void *wtz_pool_alloc_global(wtz_pool_t *p, unsigned long long n){
	n = (n + 255ull) & ~255ull;          /* the global cursor moves in 256-byte steps: slab offsets fit the tag */
#if defined(__HIP_DEVICE_COMPILE__)
	const unsigned long long o = atomicAdd(&p->used, n);
#else
	const unsigned long long o = p->used; p->used += n;
#endif
	if(o + n > p->cap){ p->overflow = 1; return NULL; }
	return p->base + o;
}

WTZ_HD void *wtz_pool_alloc(wtz_pool_t *p, size_t bytes){
	unsigned long long n = ((unsigned long long)bytes + 15ull) & ~15ull;
	/* a pool that has refused a request stays refusing until its reset (the stage ends in WTZ_E_POOL whatever else it computes): no later request moves a shard
	 * cursor, so the 32-bit offset of a shard that lost its slab cannot be driven round into the tag by the requests of a failing stage */
	if(p->overflow) return NULL;
	if(p->fail_at){
#if defined(__HIP_DEVICE_COMPILE__)
		const unsigned int k = atomicAdd(&p->nalloc, 1u);
#else
		const unsigned int k = p->nalloc++;
#endif
		if(k + 1u >= p->fail_at){ p->overflow = 1; return NULL; }
	}
#if defined(__HIP_DEVICE_COMPILE__)
	const unsigned long long slab = p->slab;
	if(slab == 0 || n > slab / 4) return wtz_pool_alloc_global(p, n);
	unsigned long long *cur = &p->shard[(blockIdx.x & (unsigned)(WTZ_POOL_NSHARD - 1)) * 16u];
	const unsigned long long st = atomicAdd(cur, n);
	const unsigned long long tag = st >> 32, off = st & 0xFFFFFFFFull;
	if(off + n <= slab) return p->base + ((tag - 1ull) << 8) + off;            /* inside the shard's slab (a shard without a slab reads as taken to the end) */
	if(off <= slab){
		/* this request crossed the end of the slab: it is the one that fetches the next */
		const unsigned long long o = atomicAdd(&p->used, slab);
		if(o + slab > p->cap){
			/* the tail of the pool: no further slab.  The shard is parked (tag 0, offset just beyond a slab: every later request of the shard goes to the global
			 * cursor, which is beyond the pool's end by now and refuses - see the test at the top) */
			atomicExch(cur, slab + 1ull);
			if(o + n <= p->cap) return p->base + o;                                /* the tail still holds this request */
			p->overflow = 1; return NULL;
		}
		atomicExch(cur, (((o >> 8) + 1ull) << 32) | n);
		return p->base + o;
	}
	return wtz_pool_alloc_global(p, n);                                         /* between the crossing and the publication of the next slab */
#else
	return wtz_pool_alloc_global(p, n);
#endif
}

/* Pointers into the pool are generic (flat) by type.  A flat store inside a loop that also reads LDS forces the compiler to
 * wait for the store to COMPLETE before the next LDS access (a flat address might be LDS), i.e. one HBM write latency per DP
 * row.  Stores through an explicit global-address-space pointer have no such ordering with LDS. */
#if defined(__HIP_DEVICE_COMPILE__)
#define WTZ_GLOBAL_AS __attribute__((address_space(1)))
template<typename T> WTZ_D WTZ_GLOBAL_AS T *wtz_as_global(T *p){ return (WTZ_GLOBAL_AS T*)p; }
#define WTZ_LDS_AS __attribute__((address_space(3)))
template<typename T> WTZ_D WTZ_LDS_AS T *wtz_as_lds(T *p){ return (WTZ_LDS_AS T*)p; }
#elif defined(__HIPCC__)
#define WTZ_GLOBAL_AS
template<typename T> __host__ __device__ static inline T *wtz_as_global(T *p){ return p; }
#define WTZ_LDS_AS
template<typename T> __host__ __device__ static inline T *wtz_as_lds(T *p){ return p; }
#endif

/* ---------------- wave-cooperative helpers ----------------
 * Tasks launched with wtz_launch_coop run with all 64 lanes of one wavefront; the host emulation runs them with
 * one lane.  Uniform values (pointers, counts) are produced by lane 0 and broadcast. */
#if defined(__HIP_DEVICE_COMPILE__)
#define WTZ_LANE ((uint32_t)(threadIdx.x & 63u))
#define WTZ_NLANES 64u
WTZ_D uint32_t wtz_coop_excl_scan(uint32_t v, uint32_t *total){
	uint32_t x = v; const uint32_t lane = WTZ_LANE;
	#pragma unroll
	for(int d = 1; d < 64; d <<= 1){ uint32_t y = __shfl_up(x, d, 64); if(lane >= (uint32_t)d) x += y; }
	*total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);      /* v_readlane: known uniform to the compiler (a shuffle's result is not, and branches on it become masked regions) */
	return x - v;
}
WTZ_D uint64_t wtz_coop_bcast64(uint64_t v){      /* two v_readfirstlane (like wtz_coop_bcast32: every caller runs with all lanes active): the result is KNOWN uniform to the compiler, a shuffle's is not */
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
	return ((uint64_t)hi << 32) | lo;
}
WTZ_D uint32_t wtz_coop_bcast32(uint32_t v){ return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
/* rank of this lane among the lanes whose predicate holds (ballot + mbcnt: no cross-lane data movement), and their number */
WTZ_D uint32_t wtz_coop_rank(bool keep, uint32_t *total){
	const unsigned long long m = __ballot(keep);
	*total = (uint32_t)__popcll(m);
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
/* bitonic sort of one u64 key per lane across the wavefront (ascending by lane), registers + ds_bpermute only */
WTZ_D uint64_t wtz_wave_sort64(uint64_t v){
	const uint32_t lane = WTZ_LANE;
	#pragma unroll
	for(uint32_t k = 2; k <= 64; k <<= 1){
		#pragma unroll
		for(uint32_t j = k >> 1; j > 0; j >>= 1){
			const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)v, (int)j, 64), ohi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), (int)j, 64);
			const uint64_t o = ((uint64_t)ohi << 32) | olo;
			const bool take_min = (((lane & k) == 0) == ((lane & j) == 0));
			const uint64_t mn = v < o ? v : o, mx = v < o ? o : v;
			v = take_min ? mn : mx;
		}
	}
	return v;
}
/* the same network over one u32 key per lane (half the cross-lane traffic of the 64-bit form) */
WTZ_D uint32_t wtz_wave_sort32(uint32_t v){
	const uint32_t lane = WTZ_LANE;
	#pragma unroll
	for(uint32_t k = 2; k <= 64; k <<= 1){
		#pragma unroll
		for(uint32_t j = k >> 1; j > 0; j >>= 1){
			const uint32_t o = (uint32_t)__shfl_xor((int)v, (int)j, 64);
			const bool take_min = (((lane & k) == 0) == ((lane & j) == 0));
			const uint32_t mn = v < o ? v : o, mx = v < o ? o : v;
			v = take_min ? mn : mx;
		}
	}
	return v;
}
/* minimum over all lanes, uniform */
WTZ_D uint32_t wtz_coop_min32(uint32_t v){
	for(int d = 32; d > 0; d >>= 1){ const uint32_t y = (uint32_t)__shfl_xor((int)v, d, 64); v = y < v ? y : v; }
	return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);      /* every lane holds the minimum: tell the compiler it is uniform */
}
/* inclusive running maximum over the lanes */
WTZ_D uint32_t wtz_coop_incl_max32(uint32_t v){
	const uint32_t lane = WTZ_LANE;
	#pragma unroll
	for(int d = 1; d < 64; d <<= 1){ const uint32_t y = (uint32_t)__shfl_up((int)v, d, 64); if(lane >= (uint32_t)d) v = y > v ? y : v; }
	return v;
}
WTZ_D unsigned long long wtz_coop_ballot(bool p){ return __ballot(p); }
/* value of lane `src` (any lane per lane) */
WTZ_D uint32_t wtz_coop_shfl32(uint32_t v, uint32_t src){ return (uint32_t)__shfl((int)v, (int)src, 64); }
/* value of lane `l` (l uniform) */
WTZ_D uint32_t wtz_coop_lane32(uint32_t v, uint32_t l){ return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane((int)l)); }
#define WTZ_WAVE_SYNC() __threadfence_block()
#else
/* single-lane forms for the host emulation; under hipcc's host pass they must also be callable from the (unused) host
 * instantiation of device functions */
#if defined(__HIPCC__)
#define WTZ_COOP_HOST __host__ __device__ static inline
#else
#define WTZ_COOP_HOST static inline
#endif
#define WTZ_LANE 0u
#define WTZ_NLANES 1u
WTZ_COOP_HOST uint32_t wtz_coop_excl_scan(uint32_t v, uint32_t *total){ *total = v; return 0; }
WTZ_COOP_HOST uint64_t wtz_coop_bcast64(uint64_t v){ return v; }
WTZ_COOP_HOST uint32_t wtz_coop_bcast32(uint32_t v){ return v; }
WTZ_COOP_HOST uint32_t wtz_coop_rank(bool keep, uint32_t *total){ *total = keep ? 1u : 0u; return 0; }
WTZ_COOP_HOST uint32_t wtz_coop_lane32(uint32_t v, uint32_t){ return v; }
WTZ_COOP_HOST uint32_t wtz_coop_min32(uint32_t v){ return v; }
WTZ_COOP_HOST uint32_t wtz_coop_incl_max32(uint32_t v){ return v; }
WTZ_COOP_HOST unsigned long long wtz_coop_ballot(bool p){ return p ? 1ull : 0ull; }
WTZ_COOP_HOST uint32_t wtz_coop_shfl32(uint32_t v, uint32_t){ return v; }
#define WTZ_WAVE_SYNC() do {} while(0)
#endif

/* phase profiler, compiled in only with -DWTZ_PROFILE: shader-clock ticks / event counts per slot, accumulated by lane 0 of each task
 * in an LDS array of the workgroup (global same-address atomics per event made the profiled kernels 3x slower) and added to the
 * global slots once, when the task ends (WTZ_PROF_BEGIN / WTZ_PROF_END in the kernel wrappers); WTZ_PROFILE_PAIR=1 prints them */
#if defined(__HIPCC__) && defined(WTZ_PROFILE)
__device__ unsigned long long wtz_prof[64];
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ unsigned long long *wtz_prof_lds(){ __shared__ unsigned long long a[64]; return a; }
#define WTZ_PROF_BEGIN() do { if(threadIdx.x < 64) wtz_prof_lds()[threadIdx.x] = 0; __syncthreads(); } while(0)
#define WTZ_PROF_END() do { __syncthreads(); if(threadIdx.x < 64){ const unsigned long long v_ = wtz_prof_lds()[threadIdx.x]; if(v_){ if(threadIdx.x == 15 || threadIdx.x == 34 || threadIdx.x == 37 || threadIdx.x == 40 || threadIdx.x == 42 || threadIdx.x == 45) atomicMax(&wtz_prof[threadIdx.x], v_); else atomicAdd(&wtz_prof[threadIdx.x], v_); } } } while(0)
#else
#define WTZ_PROF_BEGIN() do { } while(0)
#define WTZ_PROF_END() do { } while(0)
static __host__ __device__ inline unsigned long long *wtz_prof_lds(){ return NULL; }
#endif
#define WTZ_PROF_T() ((unsigned long long)clock64())
#define WTZ_PROF_ADD(slot, t0) do { if(WTZ_LANE == 0) atomicAdd(&wtz_prof_lds()[slot], (unsigned long long)clock64() - (t0)); } while(0)
#define WTZ_PROF_CNT(slot, v) do { if(WTZ_LANE == 0) atomicAdd(&wtz_prof_lds()[slot], (unsigned long long)(v)); } while(0)
#define WTZ_PROF_MAX(slot, t0) do { if(WTZ_LANE == 0) atomicMax(&wtz_prof_lds()[slot], (unsigned long long)clock64() - (t0)); } while(0)
#else
#define WTZ_PROF_BEGIN() do { } while(0)
#define WTZ_PROF_END() do { } while(0)
#define WTZ_PROF_T() 0ull
#define WTZ_PROF_ADD(slot, t0) do { (void)(t0); } while(0)
#define WTZ_PROF_CNT(slot, v) do { } while(0)
#define WTZ_PROF_MAX(slot, t0) do { (void)(t0); } while(0)
#endif

/* phase clock of the seed-lookup workgroups, compiled in only with -DWTZ_PROFILE_CAND: ticks per phase, added to global slots by thread 0 of the workgroup
 * (one atomic per phase per query); printed with the counters when WTZ_PROFILE_PAIR=1 */
#if defined(__HIPCC__) && defined(WTZ_PROFILE_CAND)
__device__ unsigned long long wtz_prof_cand[16];
#define WTZ_CPROF_T() ((unsigned long long)clock64())
#define WTZ_CPROF_ADD(slot, t0) do { if(threadIdx.x == 0){ const unsigned long long n_ = (unsigned long long)clock64(); atomicAdd(&wtz_prof_cand[slot], n_ - (t0)); (t0) = n_; } } while(0)
#define WTZ_CPROF_CNT(slot, v) do { if(threadIdx.x == 0) atomicAdd(&wtz_prof_cand[slot], (unsigned long long)(v)); } while(0)
#else
#define WTZ_CPROF_T() 0ull
#define WTZ_CPROF_ADD(slot, t0) do { (void)(t0); } while(0)
#define WTZ_CPROF_CNT(slot, v) do { } while(0)
#endif

/* growable vector living in the pool (old storage is simply abandoned on growth) */
template<typename T> struct wtz_vec {
	T *a; uint32_t n, cap; wtz_pool_t *pool; int bad;
	WTZ_HDM void init(wtz_pool_t *p, uint32_t c){ pool = p; n = 0; cap = 0; a = NULL; bad = 0; if(c) reserve(c); }
	WTZ_HDM bool reserve(uint32_t want){
		if(want <= cap) return true;
		uint32_t c = cap ? cap : 16; while(c < want) c <<= 1;
		T *b = (T*)wtz_pool_alloc(pool, (size_t)c * sizeof(T));
		if(b == NULL){ bad = 1; return false; }
		{
			/* old and new storage never overlap: batches of loads, then stores (element by element every store waits for the load behind it) */
			const T *__restrict__ src = a; T *__restrict__ dst = b;
			constexpr uint32_t B = sizeof(T) <= 4 ? 4 : (sizeof(T) <= 8 ? 2 : 1);
			uint32_t i = 0;
			if(B > 1) for(; i + B <= n; i += B){ T t[B]; for(uint32_t k = 0; k < B; k++) t[k] = src[i + k]; for(uint32_t k = 0; k < B; k++) dst[i + k] = t[k]; }
			for(; i < n; i++) dst[i] = src[i];
		}
		a = b; cap = c; return true;
	}
	WTZ_HDM bool push(const T &x){ if(n == cap && !reserve(n + 1)) return false; a[n++] = x; return true; }
};

/*
 * Exact restatement of the reference's unstable sort (sort.h:104-155): its tie order is
 * observable in wtzmo's output (SURVEY §8a trap 1), so the product's kernels run the same
 * swap sequence.  GT is a functor: gt(a,b) != 0 iff "a is greater than b" as written at the
 * reference call site.
 */
template<typename T, typename GT>
WTZ_HD void wtz_sort_exact(T *v, size_t n, GT gt){
	if(n < 2) return;
	uint32_t lo_stk[64], hi_stk[64]; int sp = 0;
	T piv, tmp;
	lo_stk[sp] = 0; hi_stk[sp] = (uint32_t)(n - 1); sp++;
	while(sp){
		sp--;
		size_t s = lo_stk[sp], e = hi_stk[sp], m = s + (e - s) / 2;
		if(gt(v[s], v[m])){ tmp = v[s]; v[s] = v[m]; v[m] = tmp; }
		if(gt(v[m], v[e])){
			tmp = v[e]; v[e] = v[m]; v[m] = tmp;
			if(gt(v[s], v[m])){ tmp = v[s]; v[s] = v[m]; v[m] = tmp; }
		}
		piv = v[m];
		size_t i = s + 1, j = e - 1;
		for(;;){
			while(gt(piv, v[i])) i++;
			while(gt(v[j], piv)) j--;
			if(i < j){ tmp = v[i]; v[i] = v[j]; v[j] = tmp; i++; j--; }
			else break;
		}
		if(i == j){ i++; j--; }
		if(j - s > e - i){
			if(s + 4 < j){ lo_stk[sp] = (uint32_t)s; hi_stk[sp] = (uint32_t)j; sp++; }
			if(i + 4 < e){ lo_stk[sp] = (uint32_t)i; hi_stk[sp] = (uint32_t)e; sp++; }
		} else {
			if(i + 4 < e){ lo_stk[sp] = (uint32_t)i; hi_stk[sp] = (uint32_t)e; sp++; }
			if(s + 4 < j){ lo_stk[sp] = (uint32_t)s; hi_stk[sp] = (uint32_t)j; sp++; }
		}
	}
	for(size_t i = 0; i < n; i++){
		bool swapped = false;
		for(size_t j = n - 1; j > i; j--){
			if(gt(v[j-1], v[j])){ tmp = v[j-1]; v[j-1] = v[j]; v[j] = tmp; swapped = true; }
		}
		if(!swapped) break;
	}
}

/* reference binary heap (list.h:78-144); cmp(a,b) returns <0,0,>0 */
template<typename T, typename CMP>
WTZ_HD void wtz_heap_push(T *h, uint32_t &n, T x, CMP cmp){
	uint32_t i = n++; h[i] = x;
	while(i){ uint32_t p = (i - 1) >> 1; if(cmp(h[i], h[p]) >= 0) break; T t = h[i]; h[i] = h[p]; h[p] = t; i = p; }
}
template<typename T, typename CMP>
WTZ_HD void wtz_heap_sift(T *h, uint32_t n, uint32_t idx, CMP cmp){
	while((idx << 1) + 1 < n){
		uint32_t pick = idx, l = (idx << 1) + 1, r = l + 1;
		if(cmp(h[pick], h[l]) > 0) pick = l;
		if(r < n && cmp(h[pick], h[r]) > 0) pick = r;
		if(pick == idx) break;
		T t = h[idx]; h[idx] = h[pick]; h[pick] = t; idx = pick;
	}
}

#endif
