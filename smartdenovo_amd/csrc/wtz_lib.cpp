/*
 * wtz_lib.cpp — libwtzmo_hip.so: kernels + stage orchestration + C ABI (include/wtzmo_hip.h).
 *
 * Built with:  hipcc -x hip --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC wtz_lib.cpp
 * (tests/emul/ also compiles this file with g++ -DWTZ_EMUL to run every "kernel" as a host loop; that
 *  build is a debugging aid for containers without a GPU and is never part of the product.)
 *
 * Execution model of this first path: every stage is a flat grid of independent tasks (one lane per
 * read / query / pair / window), 64-thread workgroups so that each wave is scheduled on its own and
 * the >= thousands of waves per launch spread over all 256 CUs / 8 XCDs; all per-task scratch and
 * results are carved from one HBM bump pool (wtz_pool_t) with a single 64-bit atomic per allocation.
 * The wave-parallel banded-DP kernels (wtz_sw_wave.h) replace the scalar K-sw3 body in wtz_pairs_align.
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <vector>
#include <atomic>
#include <algorithm>

#include "wtz_tasks.h"
#include "wtz_sw_frame.h"
#include "wtz_stitch_fused.h"

/* ------------------------------------------------------------------------------------------------ */
/* device abstraction                                                                               */
/* ------------------------------------------------------------------------------------------------ */
static thread_local char g_err[512] = "";
#define CHK(call) do { int rc_ = (call); if(rc_ != WTZ_OK) return rc_; } while(0)
#define CHK0(call) CHK(call)
#define STAGE(c, name) do { if((c)->env_trace){ (void)dev_sync(); fprintf(stderr, "[stage] %s\n", name); fflush(stderr); } } while(0)
static int wtz_fail(int code, const char *fmt, ...){
	va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
	return code;
}

#ifndef WTZ_CAND_LDS_BYTES
#define WTZ_CAND_LDS_BYTES 16384u     /* LDS window of the candidate-tuple sort */
#endif
static_assert((WTZ_CAND_LDS_BYTES & (WTZ_CAND_LDS_BYTES - 1u)) == 0u && WTZ_CAND_LDS_BYTES >= 4096u, "the windowed bitonic sort of K_candidates needs a power-of-two LDS window");
#ifndef WTZ_PAIR_DM_LDS_TIER2
#define WTZ_PAIR_DM_LDS_TIER2 49152u
#endif
/* the last two launches keep only the band work arrays in LDS (the per-match image of a strand goes to the pool when it does not fit):
 * what a heavy pair needs is resident waves, not LDS - a 159 KB slice meant ONE wave per CU, and the repeat-rich 40 Mbp set spent 14 of
 * its 15 s there (159 KB: 13.8 s, 76: 7.1, 50: 5.0, 36: 4.1).  Tier 3 = eight waves per CU with room for ~2 000 linear groups per
 * strand, tier 4 = four waves per CU with 8 191. */
#ifndef WTZ_PAIR_DM_LDS_TIER3
#define WTZ_PAIR_DM_LDS_TIER3 20480u
#endif
#ifndef WTZ_PAIR_DM_LDS_TIER4
#define WTZ_PAIR_DM_LDS_TIER4 36864u
#endif
/* kernel name tags (rocprofv3 shows wtz_kernel_*<K_pair, ...>) */
struct K_candidates;
struct K_candidates_stream;
struct K_candidates_wg;
struct K_extjob_scalar;
struct K_cigar_text;
struct K_misc;
struct K_gap;
struct K_gap_wide;
struct K_kcount;
struct K_lcount;
struct K_lplan;
struct K_ldp;
struct K_ltb;
struct K_lfold;
struct K_gplan;
struct K_gdp;
struct K_gtb;
struct K_glist;
struct K_poolalloc;
struct K_kfill;
struct K_kinsert;
struct K_khead;
struct K_kdistinct;
struct K_kinsert_total;
struct K_pack_groups;
struct K_kstats;
struct K_pack_cigars;
struct K_pack_windows;
struct K_pair;
struct K_pair_dm;
struct K_pair_zbig;
struct K_pair_big;
struct K_refine;
struct K_pack_ascii;
struct K_pack_fix;
struct K_revcomp_views;
struct K_stitch_fin;
struct K_stitch_left;
struct K_stitch_mid;
struct K_winalign;
struct K_winalign4;
struct K_winalign_big;
struct K_zfill; struct K_zread;
struct K_zrun;
struct K_zdistinct;
struct K_zdn;
struct K_zcount;

#include <chrono>
static double wtz_wall(){ return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

#ifndef WTZ_EMUL
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#define WTZ_LAMBDA __device__
#ifndef WTZ_OCC_WINALIGN
#define WTZ_OCC_WINALIGN 3
#endif
/* The pair kernels are latency-bound, so waves per SIMD pay - until the register budget of the occupancy target forces spills into the loops: with the window scan of
 * round 4 (K_pair needs 135 VGPRs, K_pair_dm 149) five waves = 96 VGPRs and 76 / 42 spilled registers cost more than the fifth wave brings.  configs[2], ms per step:
 * K_pair 1 731 (5 waves) / 773 (4) / 916 (3); K_pair_dm 5 384 / 5 149 / 5 282. */
#ifndef WTZ_OCC_PAIR_DM
#define WTZ_OCC_PAIR_DM 4
#endif
#ifndef WTZ_OCC_PAIR
#define WTZ_OCC_PAIR 5
#endif
/* K_gap (K-sw2 on a wavefront): 213 registers when left alone - two waves per SIMD, where its 12 KB LDS slice lets a CU hold thirteen.  At three (168 registers, 34 spilled
 * values outside the row loop) the K-sw2 stage of a configs[2] step goes from 200 to 188 ms (round 6). */
#ifndef WTZ_OCC_GAP
#define WTZ_OCC_GAP 3
#endif

/* every context owns a non-blocking HIP stream; the API entry points make it current for the calling host thread, so that
 * two host threads can drive two contexts (two batches in flight) whose kernels and copies overlap on the device */
static thread_local hipStream_t g_stream = 0;
#define HIPCHK(call) do { hipError_t e_ = (call); if(e_ != hipSuccess) return wtz_fail(WTZ_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while(0)



/* TAG only names the kernel (rocprofv3 shows wtz_kernel_tasks<K_pair_seed, ...>) */
template<typename TAG, typename F> __global__ void __launch_bounds__(64) wtz_kernel_tasks(uint64_t n, F f){
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	WTZ_PROF_BEGIN();
	if(i < n) f(i);
	WTZ_PROF_END();
}
template<typename TAG, typename F> static int wtz_launch(hipStream_t st, uint64_t n, F f){
	if(n == 0) return WTZ_OK;
	const uint32_t bs = 64;
	uint64_t nb = (n + bs - 1) / bs;
	if(nb > 0x7FFFFFFFull) return wtz_fail(WTZ_E_ARG, "grid too large");
	(void)st; hipLaunchKernelGGL((wtz_kernel_tasks<TAG, F>), dim3((uint32_t)nb), dim3(bs), 0, g_stream, n, f);
	HIPCHK(hipGetLastError());
	return WTZ_OK;
}
/* Heavy, data-dependent tasks (whole pairs / windows / queries): one task per WAVEFRONT, executed by lane 0.  A flat
 * thread-per-task grid serialises up to 64 divergent control flows inside every wave; these tasks are chains of
 * dependent memory operations, so what hides their latency is the number of resident waves (up to 32 per CU, 8192 on
 * the chip), not the lanes of one wave.  Stages whose inner loops are regular get wave-cooperative kernels instead
 * (wtz_sw_wave.h). */
template<typename TAG, typename F> __global__ void __launch_bounds__(64) wtz_kernel_wave_tasks(uint64_t n, F f){
	const uint64_t i = blockIdx.x;
	WTZ_PROF_BEGIN();
	if(i < n && threadIdx.x == 0) f(i);
	WTZ_PROF_END();
}
/* wave-cooperative tasks: every lane of the wavefront enters the task body (WTZ_LANE / wtz_coop_* inside).
 * These kernels are latency-bound chains: resident waves per SIMD are their throughput, so a TAG can ask the register
 * allocator for a minimum occupancy (wtz_occ<TAG>::waves) instead of the 512-VGPR budget a 64-thread block would get. */
template<typename TAG> struct wtz_occ { static constexpr int waves = 1; };
template<> struct wtz_occ<K_winalign> { static constexpr int waves = WTZ_OCC_WINALIGN; };
template<> struct wtz_occ<K_pair> { static constexpr int waves = WTZ_OCC_PAIR; };
template<> struct wtz_occ<K_pair_dm> { static constexpr int waves = WTZ_OCC_PAIR_DM; };
template<> struct wtz_occ<K_pair_zbig> { static constexpr int waves = 2; };      /* both scan bodies (168 VGPRs + spills at three waves); a handful of pairs per launch, each a long dependent chain */
template<> struct wtz_occ<K_gap> { static constexpr int waves = WTZ_OCC_GAP; };
/* lane-per-problem K-sw1 (wtz_sw_lane.h): the band lives in 2 x (NC + 1) VGPRs */
#ifndef WTZ_OCC_LDP
#define WTZ_OCC_LDP 2
#endif
template<> struct wtz_occ<K_ldp> { static constexpr int waves = WTZ_OCC_LDP; };
template<> struct wtz_occ<K_gdp> { static constexpr int waves = WTZ_OCC_LDP; };
/* their tracebacks: chains of dependent loads, nothing to keep in registers */
template<> struct wtz_occ<K_ltb> { static constexpr int waves = 8; };
template<> struct wtz_occ<K_gtb> { static constexpr int waves = 8; };
template<typename TAG, typename F> __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(wtz_occ<TAG>::waves, 8))) wtz_kernel_coop_tasks(uint64_t n, F f){
	const uint64_t i = blockIdx.x;
	WTZ_PROF_BEGIN();
	if(i < n) f(i);
	WTZ_PROF_END();
}
template<typename TAG, typename F> static int wtz_launch_coop(hipStream_t st, uint64_t n, F f, uint32_t lds_bytes = WTZ_WAVE_LDS_BYTES){
	if(n == 0) return WTZ_OK;
	if(n > 0x7FFFFFFFull) return wtz_fail(WTZ_E_ARG, "grid too large");
	if(lds_bytes > 65536u){ HIPCHK(hipFuncSetAttribute((const void*)&wtz_kernel_coop_tasks<TAG, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); }      /* opt in to more than 64 KB of dynamic LDS */
	(void)st; hipLaunchKernelGGL((wtz_kernel_coop_tasks<TAG, F>), dim3((uint32_t)n), dim3(64), lds_bytes, g_stream, n, f);
	HIPCHK(hipGetLastError());
	return WTZ_OK;
}
/* one task per WORKGROUP of NT threads (wave-size multiples): every thread enters the task body (WTZ_WG_TID / WTZ_WG_SYNC inside) */
template<typename TAG, typename F> __global__ void __launch_bounds__(WTZ_CWG_THREADS) wtz_kernel_wg_tasks(uint64_t n, F f){
	const uint64_t i = blockIdx.x;
	if(i < n) f(i);
}
template<typename TAG, typename F> static int wtz_launch_wg(uint64_t n, F f, uint32_t nthreads, uint32_t lds_bytes){
	if(n == 0) return WTZ_OK;
	if(n > 0x7FFFFFFFull) return wtz_fail(WTZ_E_ARG, "grid too large");
	if(lds_bytes > 65536u){ HIPCHK(hipFuncSetAttribute((const void*)&wtz_kernel_wg_tasks<TAG, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); }
	hipLaunchKernelGGL((wtz_kernel_wg_tasks<TAG, F>), dim3((uint32_t)n), dim3(nthreads), lds_bytes, g_stream, n, f);
	HIPCHK(hipGetLastError());
	return WTZ_OK;
}
/* four tasks per wavefront: one per 16-lane group (wtz_sw_grp.h); f gets the index of the block's first task */
template<typename TAG, typename F> __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8))) wtz_kernel_grp_tasks(uint64_t n, F f){
	const uint64_t i = (uint64_t)blockIdx.x * 4;
	WTZ_PROF_BEGIN();
	if(i < n) f(i);
	WTZ_PROF_END();
}
template<typename TAG, typename F> static int wtz_launch_grp(uint64_t n, F f, uint32_t lds_bytes){
	if(n == 0) return WTZ_OK;
	const uint64_t nb = (n + 3) / 4;
	if(nb > 0x7FFFFFFFull) return wtz_fail(WTZ_E_ARG, "grid too large");
	hipLaunchKernelGGL((wtz_kernel_grp_tasks<TAG, F>), dim3((uint32_t)nb), dim3(64), lds_bytes, g_stream, n, f);
	HIPCHK(hipGetLastError());
	return WTZ_OK;
}
template<typename TAG, typename F> static int wtz_launch_wave(hipStream_t st, uint64_t n, F f){
	if(n == 0) return WTZ_OK;
	if(n > 0x7FFFFFFFull) return wtz_fail(WTZ_E_ARG, "grid too large");
	(void)st; hipLaunchKernelGGL((wtz_kernel_wave_tasks<TAG, F>), dim3((uint32_t)n), dim3(64), WTZ_WAVE_LDS_BYTES, g_stream, n, f);
	HIPCHK(hipGetLastError());
	return WTZ_OK;
}
/* transient buffers: stream-ordered allocation (no device-wide synchronisation, memory is recycled by the HIP mem pool);
 * long-lived buffers (reads, indexes, scratch pool): plain hipMalloc */
/* transient device buffers come from a per-context arena (host-side bump pointer over one persistent allocation, released
 * stack-wise when the API call returns): no hipMalloc/hipFree - and therefore no device-wide synchronisation - on the batch
 * path, which is what lets two contexts overlap.  Requests that do not fit fall back to hipMalloc and are freed at release. */
/* Overflow buffers are not given back to the driver when the call returns: they are kept (up to WTZ_ARENA_CACHE_BYTES) for the next request of about that size.  The
 * index builds of a 1.2 Gbp read set take 2.4 + 2.4 + 1.2 GB of sort buffers beyond the arena; hipFree + hipMalloc of those cost 30 ms on one box and 850 ms on
 * another (every repeat of the step), and each hipFree is a device-wide synchronisation. */
#define WTZ_ARENA_CACHE_BYTES ((size_t)8 << 30)      /* the three sort buffers of a configs[2] index build are 6 GB; what does not fit is given back at once */
struct wtz_arena { uint8_t *base; size_t cap, top; std::vector<void*> overflow; std::vector<size_t> overflow_bytes; std::vector<std::pair<void*, size_t> > cache; size_t cache_bytes; };
static thread_local wtz_arena *g_arena = NULL;
static void arena_cache_flush(wtz_arena *a);
static int dev_alloc(void **p, size_t n){
	n = (n + 255) & ~(size_t)255; if(n == 0) n = 256;
	if(g_arena && g_arena->top + n <= g_arena->cap){ *p = g_arena->base + g_arena->top; g_arena->top += n; return WTZ_OK; }
	if(g_arena){
		for(size_t i = 0; i < g_arena->cache.size(); i++){
			const size_t cb = g_arena->cache[i].second;
			if(cb >= n && cb <= n + n / 8 + ((size_t)1 << 20)){
				*p = g_arena->cache[i].first; g_arena->cache_bytes -= cb; g_arena->cache.erase(g_arena->cache.begin() + (long)i);
				g_arena->overflow.push_back(*p); g_arena->overflow_bytes.push_back(cb); return WTZ_OK;
			}
		}
	}
	if(hipMalloc(p, n) != hipSuccess){
		(void)hipGetLastError();
		if(g_arena && !g_arena->cache.empty()){ (void)hipDeviceSynchronize(); arena_cache_flush(g_arena); }
		HIPCHK(hipMalloc(p, n));
	}
	if(g_arena){ g_arena->overflow.push_back(*p); g_arena->overflow_bytes.push_back(n); }
	return WTZ_OK;
}
static void arena_cache_flush(wtz_arena *a){ for(size_t i = 0; i < a->cache.size(); i++) (void)hipFree(a->cache[i].first); a->cache.clear(); a->cache_bytes = 0; }
static void dev_free(void *){ /* released by the arena scope of the API call */ }
struct wtz_arena_scope { wtz_arena *a, *prev; size_t mark; size_t nover;
	wtz_arena_scope(wtz_arena *ar) : a(ar), prev(g_arena), mark(ar ? ar->top : 0), nover(ar ? ar->overflow.size() : 0) { g_arena = ar; }
	~wtz_arena_scope(){
		g_arena = prev;      /* never left pointing at an arena whose call has returned (its context may be destroyed next; wtz_ctx_destroy itself runs inside a scope) */
		if(!a) return;
		if(a->overflow.size() > nover){
			(void)hipStreamSynchronize(g_stream);
			while(a->overflow.size() > nover){
				void *q = a->overflow.back(); const size_t qb = a->overflow_bytes.back(); a->overflow.pop_back(); a->overflow_bytes.pop_back();
				if(a->cache_bytes + qb <= WTZ_ARENA_CACHE_BYTES && a->cache.size() < 16){ a->cache.push_back(std::make_pair(q, qb)); a->cache_bytes += qb; }
				else (void)hipFree(q);
			}
		}
		a->top = mark;
	} };
static int dev_alloc_persist(void **p, size_t n){
	if(hipMalloc(p, n ? n : 16) == hipSuccess) return WTZ_OK;
	(void)hipGetLastError();
	if(g_arena && !g_arena->cache.empty()){ (void)hipDeviceSynchronize(); arena_cache_flush(g_arena); }      /* the kept overflow buffers go first */
	HIPCHK(hipMalloc(p, n ? n : 16)); return WTZ_OK;
}
static void dev_free_persist(void *p){ if(p) (void)hipFree(p); }
static int dev_h2d(void *d, const void *h, size_t n){ if(n){ HIPCHK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, g_stream)); HIPCHK(hipStreamSynchronize(g_stream)); } return WTZ_OK; }
static int dev_d2h(void *h, const void *d, size_t n){ if(n){ HIPCHK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, g_stream)); HIPCHK(hipStreamSynchronize(g_stream)); } return WTZ_OK; }
static int dev_set(void *d, int v, size_t n){ if(n) HIPCHK(hipMemsetAsync(d, v, n, g_stream)); return WTZ_OK; }
static int dev_d2d(void *d, const void *s, size_t n){ if(n) HIPCHK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, g_stream)); return WTZ_OK; }
static int dev_sync(){ HIPCHK(hipStreamSynchronize(g_stream)); return WTZ_OK; }

struct wtz_timer { hipEvent_t a, b; bool ok;
	wtz_timer(){ ok = hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess; }
	~wtz_timer(){ if(ok){ (void)hipEventDestroy(a); (void)hipEventDestroy(b); } }
	void start(){ if(ok) (void)hipEventRecord(a, g_stream); }
	double stop(){ float ms = 0; if(ok){ (void)hipEventRecord(b, g_stream); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b); } return ms; }
	void lap(){ if(ok) (void)hipEventRecord(b, g_stream); }                 /* end mark now, read later */
	double read(){ float ms = 0; if(ok){ (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b); } return ms; } };

static int dev_sort_pairs_u64_u32(uint64_t *keys, uint32_t *vals, uint64_t n, unsigned end_bit){
	if(n < 2) return WTZ_OK;
	uint64_t *k2 = NULL; uint32_t *v2 = NULL; void *tmp = NULL; size_t tmp_bytes = 0; int rc;
	if((rc = dev_alloc((void**)&k2, n * 8))) return rc;
	if((rc = dev_alloc((void**)&v2, n * 4))){ dev_free(k2); return rc; }
	rocprim::double_buffer<uint64_t> kb(keys, k2); rocprim::double_buffer<uint32_t> vb(vals, v2);
	hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, kb, vb, (size_t)n, 0u, end_bit, g_stream);
	if(e == hipSuccess && (rc = dev_alloc(&tmp, tmp_bytes)) == WTZ_OK){
		e = rocprim::radix_sort_pairs(tmp, tmp_bytes, kb, vb, (size_t)n, 0u, end_bit, g_stream);
		if(e == hipSuccess) e = hipStreamSynchronize(g_stream);
		if(e == hipSuccess && kb.current() != keys){ e = hipMemcpyAsync(keys, kb.current(), n * 8, hipMemcpyDeviceToDevice, g_stream); if(e == hipSuccess) e = hipMemcpyAsync(vals, vb.current(), n * 4, hipMemcpyDeviceToDevice, g_stream); if(e == hipSuccess) e = hipStreamSynchronize(g_stream); }
	}
	dev_free(tmp); dev_free(k2); dev_free(v2);
	if(e != hipSuccess) return wtz_fail(WTZ_E_HIP, "radix_sort_pairs failed: %s", hipGetErrorString(e));
	return rc;
}
static int dev_exclusive_scan_u32(const uint32_t *in, uint32_t *out, uint64_t n){
	if(n == 0) return WTZ_OK;
	void *tmp = NULL; size_t tmp_bytes = 0;
	hipError_t e = rocprim::exclusive_scan(tmp, tmp_bytes, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), g_stream);
	if(e != hipSuccess) return wtz_fail(WTZ_E_HIP, "exclusive_scan failed: %s", hipGetErrorString(e));
	CHK0(dev_alloc(&tmp, tmp_bytes));
	e = rocprim::exclusive_scan(tmp, tmp_bytes, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), g_stream);
	if(e == hipSuccess) e = hipStreamSynchronize(g_stream);
	dev_free(tmp);
	if(e != hipSuccess) return wtz_fail(WTZ_E_HIP, "exclusive_scan failed: %s", hipGetErrorString(e));
	return WTZ_OK;
}
#else  /* ---------------- host emulation of the launch geometry (tests only) ---------------- */
#define WTZ_LAMBDA
typedef int hipStream_t;
template<typename TAG, typename F> static int wtz_launch(hipStream_t, uint64_t n, F f){ for(uint64_t i = 0; i < n; i++) f(i); return WTZ_OK; }
template<typename TAG, typename F> static int wtz_launch_wave(hipStream_t st, uint64_t n, F f){ return wtz_launch<TAG>(st, n, f); }
template<typename TAG, typename F> static int wtz_launch_coop(hipStream_t st, uint64_t n, F f, uint32_t = 0){ return wtz_launch<TAG>(st, n, f); }
template<typename TAG, typename F> static int wtz_launch_wg(uint64_t n, F f, uint32_t, uint32_t){ for(uint64_t i = 0; i < n; i++) f(i); return WTZ_OK; }
static int dev_alloc(void **p, size_t n){ *p = malloc(n ? n : 16); return *p ? WTZ_OK : wtz_fail(WTZ_E_HIP, "malloc(%zu) failed", n); }
static void dev_free(void *p){ free(p); }
static int dev_alloc_persist(void **p, size_t n){ return dev_alloc(p, n); }
static void dev_free_persist(void *p){ free(p); }
struct wtz_arena { int unused; };
struct wtz_arena_scope { std::vector<void*> *keep; wtz_arena_scope(wtz_arena*){ } };
static int dev_h2d(void *d, const void *h, size_t n){ if(n) memcpy(d, h, n); return WTZ_OK; }
static int dev_d2h(void *h, const void *d, size_t n){ if(n) memcpy(h, d, n); return WTZ_OK; }
static int dev_set(void *d, int v, size_t n){ if(n) memset(d, v, n); return WTZ_OK; }
static int dev_d2d(void *d, const void *s, size_t n){ if(n) memcpy(d, s, n); return WTZ_OK; }
static int dev_sync(){ return WTZ_OK; }
#include <time.h>
struct wtz_timer { struct timespec t0; void start(){ clock_gettime(CLOCK_MONOTONIC, &t0); }
	double stop(){ struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1); return 1e3 * (double)(t1.tv_sec - t0.tv_sec) + 1e-6 * (double)(t1.tv_nsec - t0.tv_nsec); }
	double lapv = 0; void lap(){ lapv = stop(); } double read(){ return lapv; } };
static int dev_exclusive_scan_u32(const uint32_t *in, uint32_t *out, uint64_t n){ uint32_t a = 0; for(uint64_t i = 0; i < n; i++){ uint32_t v = in[i]; out[i] = a; a += v; } return WTZ_OK; }
static int dev_sort_pairs_u64_u32(uint64_t *keys, uint32_t *vals, uint64_t n, unsigned){
	std::vector<std::pair<uint64_t, uint32_t> > v((size_t)n);
	for(uint64_t i = 0; i < n; i++) v[(size_t)i] = std::make_pair(keys[i], vals[i]);
	std::stable_sort(v.begin(), v.end(), [](const std::pair<uint64_t, uint32_t> &a, const std::pair<uint64_t, uint32_t> &b){ return a.first < b.first; });
	for(uint64_t i = 0; i < n; i++){ keys[i] = v[(size_t)i].first; vals[i] = v[(size_t)i].second; }
	return WTZ_OK;
}
#endif





/* ------------------------------------------------------------------------------------------------ */
/* context                                                                                          */
/* ------------------------------------------------------------------------------------------------ */
struct wtz_ctx {
	int device;
	/* z-index arrays of the previous build, kept for the rebuild (same read set -> same sizes): freeing and re-allocating 19 GB per
	 * build cost 0.1 - 0.8 s of hipFree / hipMalloc at configs[2], depending on what else the host's memory manager was doing */
	/* two z-index slots: [0] = the index proper (all reads, or the subset of the batch in flight), [1] = optional second index holding only the
	 * QUERIES of the batch in flight (wtz_zindex_build_queries): with several GPUs every device keeps slot 0 for its own share of the candidate
	 * reads and rebuilds the small slot 1 per batch, instead of all devices building the z-index of all reads */
	struct zslot_t {
		std::vector<std::pair<void*, size_t> > parked, live;
		uint64_t *zoff = NULL; uint64_t n_z = 0; wtz_zindex_t Z; bool have = false; bool sub = false; uint64_t sub_cap = 0;      /* sub: the slot holds a subset of the reads (rebuilt per batch) */
		zslot_t(){ memset(&Z, 0, sizeof Z); }
	} zs[2];
#ifndef WTZ_EMUL
	hipStream_t stream;
	hipStream_t stream_mw = 0; hipEvent_t ev_mw_fork = 0, ev_mw_join = 0;      /* side stream of the multi-wave K-sw3 launch */
	hipStream_t stream_gap = 0; hipEvent_t ev_gap_fork = 0, ev_gap_join = 0;   /* side stream of K_gap (runs beside the left extensions) */
#endif
	bool shares_indexes;      /* clone: reads / k-mer table / z-index belong to the parent context */
	void *kpark_p[2]; size_t kpark_b[2], klive_b[2];      /* k-mer table [0] / seed list [1]: buffer parked for the next build, size of the live one (0: not from kalloc) */
	wtz_arena arena;          /* transient device buffers of the API call in progress */
	uint32_t cap_pairs, cap_items;     /* grow-only capacity of the per-batch result arrays */
	wtz_params_t P; wtz_params_t *dP;
	/* reads */
	uint64_t *bits; uint64_t n_words; uint64_t *rdoff; uint32_t *rdlen; uint32_t n_reads;
	std::vector<uint32_t> h_rdlen;
	/* k-mer index */
	wtz_kslot_t *ktab; uint64_t kmask; uint32_t *kseeds; uint64_t n_kocc;
	/* sharded index build in progress (wtz_index_count ... wtz_index_finish): sorted occurrences and the shard's distinct k-mers */
	uint64_t *pend_keys = NULL; uint32_t *pend_vals = NULL; uint64_t pend_tot = 0; uint64_t *pend_dk = NULL, *pend_dstart = NULL; uint32_t *pend_dc = NULL; uint64_t pend_nd = 0; uint32_t pend_beg = 0, pend_end = 0;
	uint64_t *cq_gptr = NULL; uint32_t cq_gcap = 0; bool cq_groups = false; std::vector<uint32_t> cq_ng;
	uint32_t idx_beg = 0, idx_end = 0; bool idx_len_sorted = false;      /* read range of the k-mer index; lengths non-increasing inside it (true unless -b clipped reads after the sort) */
	/* z index */
	/* pool */
	/* scratch: ONE allocation of pool_bytes, cut in two bump pools: dpool[0] = results and scratch that live for the batch (match lists,
	 * windows, anchors, CIGARs), dpool[1] = the transient pool of the K-sw3 trace matrices, reset after every launch group of extension
	 * jobs (run_extjobs sizes the groups from the jobs' geometry, so its demand is planned, not discovered by exhaustion) */
	wtz_pool_t *dpool; uint8_t *pool_base; uint64_t pool_bytes, main_bytes;
	/* per-batch results */
	uint32_t *d_qid, *d_cid; wtz_pairres_t *d_pairres; uint32_t n_pairs; std::vector<wtz_pairres_t> h_pairres;
	wtz_alnres_dev_t *d_alnres; uint32_t n_items; std::vector<wtz_alnres_dev_t> h_alnres;
	char *d_text = NULL; size_t cap_text = 0;      /* rendered CIGAR text of the last alignment call (wtz_fetch_cigar_text / wtz_cigar_text_device) */
#ifndef WTZ_EMUL
	/* copies are numbered: copy k signals ev_text_done[k & 1].  text_begun = copies started (device-stage thread), text_known_done = copies known to have finished
	 * (a render waits for the latest one before it refills d_text), text_ended = copies whose end has been asked for (the caller's commit thread, in the same order).
	 * Round 5 had ONE event and a plain bool shared by the two threads: _end could wait on the event after _begin had re-recorded it for the next range. */
	hipStream_t stream_copy = 0; hipEvent_t ev_text_ready = 0, ev_text_done[2] = {0, 0}; std::atomic<uint64_t> text_begun{0}, text_known_done{0}, text_ended{0};      /* wtz_fetch_cigar_text_begin / _end: the text's way to the host beside the next range's kernels */
#endif
	bool have_pairs, have_items;
	/* candidate request in flight (wtz_candidates_begin / _end) */
	uint32_t *cq_thr = NULL;
	uint32_t *cq_q = NULL, *cq_nc = NULL; uint64_t *cq_cand = NULL; unsigned long long *cq_bytes = NULL; uint32_t cq_cap = 0, cq_n = 0; bool cq_pending = false; wtz_timer cq_tm;
	wtz_counters_t cnt;
	uint64_t tpool_peak_call = 0, main_used_call = 0;      /* transient-pool high-water mark / main-pool bytes of the API call in progress */
	uint32_t env_xcd_group = 256;   /* WTZ_XCD_GROUP: consecutive pairs per XCD run in K_pair (0 = identity block -> pair mapping) */
	int env_zread = 1;           /* WTZ_ZREAD=0: every read's z-mer index by the device-wide form (strided fill + radix sort) instead of one workgroup per read (wtz_task_zread) */
	int env_ext_fused = 1;       /* WTZ_EXT_FUSED=0: the two end extensions of a stitched overlap in two launches with K_stitch_mid between them instead of on one wavefront (wtz_stitch_fused.h) */
	int last_pool_fail = 0;      /* which pool the last WTZ_E_POOL came from: 1 = main, 2 = transient (wtz_pool_failure_kind) */
	double ext_use_ratio = 0.4; uint64_t tpool_last_used = 0;      /* run_stitch_fused: share of the trace upper bounds the fused launches have really taken */
	bool fused_ran = false;      /* this stitch stage's fused launch has run: the extension launches behind it only sweep up what it left open */
	int env_ext_mw_rows = 0;     /* WTZ_EXT_MW_ROWS=<n>: items whose two extensions can run at least n rows go to the four-wave frame kernel (wtz_sw_frame_mw.h) beside the fused launch.
	                              * Off by default: measured at configs[2] (gpurun_out/r06d) K-sw3 418 ms without it, 490 ms with n = 2048 (1 000 of 30 000 items per range), 654 ms with
	                              * n = 1024, 418 ms with n = 4096 (5-8 items per range) - the launches are bound by the row RATE of the resident wavefronts (time = ~5 ms + 0.65 ms per
	                              * million rows), not by their longest job, and four wavefronts spend 2.2 x the instructions of one on a row */
	int env_ext_pk = 1;          /* WTZ_EXT_PK=0: K-sw3 without the packed 16-bit form (wtz_sw_frame16.h) in front of the 32-bit frame form */
	unsigned long long ext_fr_total = 0;        /* items dealt to the 32-bit form before the launch (geometry outside the 16-bit window) */
	unsigned long long ext_open_total = 0;      /* items the packed form declined (outside its 16-bit window) and the 32-bit form finished */
	int env_ext_fr = 1;          /* WTZ_EXT_FR=0: K-sw3 one-wave jobs on the round-4 register kernel (wtz_extend_shift_reg) instead of the frame form (wtz_sw_frame.h) */
	int env_heavy_first = -1;    /* WTZ_PAIR_HEAVY_FIRST: the heaviest pairs of a K_pair launch first (-1 = engine default: dmo on, zmo off) */
	int env_cand_wg = 1;         /* WTZ_CAND_WG=0: the one-wavefront-per-query sorting form of the seed lookup (the form before round 3) */
	int env_cand_stream = 0;     /* WTZ_CAND_STREAM=1: sort-free candidate accumulation (LDS sketch + survivor table, wtz_seed.h); bit-exact, pays at 25x coverage only: see DESIGN.md */
	int env_gap_lane = 1;        /* WTZ_GAP_LANE=0: every gap on a wavefront (the form before round 3) */
	int env_lane = 1;            /* WTZ_WINALIGN_LANE=0: the chained wave-per-window kernel for every window (the form before round 3); 2: run both and compare */
	int env_grp4 = 0;            /* WTZ_WINALIGN4=1: four windows per wavefront first (wtz_sw_grp.h; bit-exact, measured 2x SLOWER than one window per wave: see DESIGN.md) */
	bool env_trace = false;      /* WTZ_STAGE_TRACE: name every device stage on stderr before it is launched (locating a device fault) */
	bool env_fail_once = false;      /* WTZ_POOL_FAIL_ONCE: the injected failure hits one stage call only (the retry must then succeed) */
	unsigned env_fail_at = 0, env_tfail_at = 0;      /* WTZ_POOL_FAIL_AT / WTZ_TPOOL_FAIL_AT: fault injection into the main / transient pool */
	int env_dm_first_big = 1; int env_sw_mode = 0, env_use_reg = 1, env_gap_side = 0; bool env_profile = false;     /* WTZ_* debugging switches, read in wtz_ctx_create */
};

#ifndef WTZ_EMUL
#define CTX_ENTER(c) (void)hipSetDevice((c)->device); g_stream = (c)->stream; wtz_arena_scope arena_scope_(&(c)->arena)
#else
#define CTX_ENTER(c) wtz_arena_scope arena_scope_(&(c)->arena)
#endif

static wtz_reads_t ctx_reads(const wtz_ctx *c){ wtz_reads_t R; R.bits = c->bits; R.rdoff = c->rdoff; R.rdlen = c->rdlen; R.n_reads = c->n_reads; return R; }
static wtz_env_t ctx_env(const wtz_ctx *c){ wtz_env_t V; V.R = ctx_reads(c); V.Z = c->zs[0].Z; V.ZQ = c->zs[1].have ? c->zs[1].Z : c->zs[0].Z; V.P = c->dP; V.pool = c->dpool; V.dm_first_big = (uint32_t)c->env_dm_first_big; return V; }

static int tpool_reset(wtz_ctx *c){
	wtz_pool_t p; wtz_pool_init(&p, c->pool_base + c->main_bytes, c->pool_bytes - c->main_bytes, c->env_tfail_at);
	return dev_h2d(c->dpool + 1, &p, sizeof p);
}
static int pool_reset(wtz_ctx *c){
	wtz_pool_t p; wtz_pool_init(&p, c->pool_base, c->main_bytes, c->env_fail_at);
	CHK(dev_h2d(c->dpool, &p, sizeof p));
	c->tpool_peak_call = 0;
	return tpool_reset(c);
}
/* the transient pool after a launch group: remember its high-water mark, fail on exhaustion */
static int tpool_check(wtz_ctx *c, const char *stage){
	wtz_pool_t p; CHK(dev_d2h(&p, c->dpool + 1, sizeof p));
	const uint64_t u = p.used > p.cap ? p.cap : p.used;
	c->tpool_last_used = u;
	if(u > c->tpool_peak_call) c->tpool_peak_call = u;
	if(p.overflow && c->env_fail_once) c->env_tfail_at = 0;
	if(p.overflow) c->last_pool_fail = 2;
	if(p.overflow) return wtz_fail(WTZ_E_POOL, "%s: transient trace pool exhausted (%llu of %llu bytes requested); use a larger pool",
		stage, (unsigned long long)p.used, (unsigned long long)p.cap);
	return WTZ_OK;
}
static int pool_check(wtz_ctx *c, const char *stage){
	wtz_pool_t p; CHK(dev_d2h(&p, c->dpool, sizeof p));
	const uint64_t u = (p.used > p.cap ? p.cap : p.used);
	c->main_used_call = u;
	if(u + c->tpool_peak_call > c->cnt.pool_peak) c->cnt.pool_peak = u + c->tpool_peak_call;
	if(p.overflow && c->env_fail_once) c->env_fail_at = 0;       /* injected failure: only the first stage call that gets that far */
	if(p.overflow) c->last_pool_fail = 1;
	if(p.overflow) return wtz_fail(WTZ_E_POOL, "%s: device scratch pool exhausted (%llu of %llu bytes requested); use fewer items per call or a larger pool",
		stage, (unsigned long long)p.used, (unsigned long long)p.cap);
	return WTZ_OK;
}

extern "C" const char *wtz_last_error(void){ return g_err; }

extern "C" int wtz_device_count(void){
#ifndef WTZ_EMUL
	int n = 0; if(hipGetDeviceCount(&n) != hipSuccess) return 0; return n;
#else
	return 1;
#endif
}

extern "C" int wtz_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes){
	if(!free_bytes || !total_bytes) return wtz_fail(WTZ_E_ARG, "null argument");
#ifndef WTZ_EMUL
	int prev = 0; HIPCHK(hipGetDevice(&prev)); HIPCHK(hipSetDevice(device));
	size_t fr = 0, tot = 0; const hipError_t e = hipMemGetInfo(&fr, &tot); (void)hipSetDevice(prev);
	if(e != hipSuccess) return wtz_fail(WTZ_E_HIP, "hipMemGetInfo: %s", hipGetErrorString(e));
	*free_bytes = fr; *total_bytes = tot;
#else
	(void)device; *free_bytes = *total_bytes = 16ull << 30;
#endif
	return WTZ_OK;
}

extern "C" int wtz_ctx_create(int device, const wtz_params_c *params, uint64_t pool_bytes, wtz_ctx_t **out){
	if(!params || !out) return wtz_fail(WTZ_E_ARG, "null argument");
	if(params->ksize < 5 || params->ksize > 32 || params->zsize < 5 || params->zsize > 16 || params->ksave < 1) return wtz_fail(WTZ_E_ARG, "k/z/S out of range (wtzmo.c:1658-1660)");
#ifndef WTZ_EMUL
	int ndev = 0;
	if(hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return wtz_fail(WTZ_E_HIP, "no HIP device visible: libwtzmo_hip needs an MI355X (gfx950); there is no CPU fallback");
	if(device < 0 || device >= ndev) return wtz_fail(WTZ_E_ARG, "device %d out of range (%d visible)", device, ndev);
	HIPCHK(hipSetDevice(device));
#endif
	wtz_ctx *c = new wtz_ctx();
	c->device = device; c->P = *params; c->dP = NULL; c->shares_indexes = false;
	c->kpark_p[0] = c->kpark_p[1] = NULL; c->kpark_b[0] = c->kpark_b[1] = 0; c->klive_b[0] = c->klive_b[1] = 0;
#ifndef WTZ_EMUL
	c->stream = 0;
#endif
	c->bits = NULL; c->rdoff = NULL; c->rdlen = NULL; c->n_reads = 0; c->n_words = 0;
	c->ktab = NULL; c->kseeds = NULL; c->kmask = 0; c->n_kocc = 0;
	c->dpool = NULL; c->pool_base = NULL; c->pool_bytes = pool_bytes ? pool_bytes : (4ull << 30);
#ifndef WTZ_EMUL
	if(!pool_bytes){
		/* default: 45 % of the free HBM, at most 128 GB (288 GB per MI355X; reads + both indexes of a 1.2 Gbp set take ~25 GB).  The host driver
		 * cuts a batch into ranges that fit the pool (wtz_pool_info) and every stage of a range ends in the tail of its slowest tasks:
		 * configs[2] runs in 36 ranges / 5.30 s with 64 GB, 22 ranges / 5.17 s with 128 GB, no further gain at 200 GB.  A second context on
		 * the same device (--workers 2, wtz_ctx_clone) takes 45 % of what is left. */
		size_t fr = 0, tot = 0;
		if(hipMemGetInfo(&fr, &tot) == hipSuccess && fr / 20 * 9 > c->pool_bytes) c->pool_bytes = (uint64_t)(fr / 20 * 9);
		if(c->pool_bytes > (128ull << 30)) c->pool_bytes = 128ull << 30;
		/* WTZ_DEFAULT_POOL_MB: the size a caller gets that did not ask for one (tests: the 128 GB default costs 4.4 s of hipMalloc per process - a hundred
		 * small golden cases spent 400 s of the GPU suite allocating; the pool size never changes a result, only the number of ranges) */
		if(const char *e = getenv("WTZ_DEFAULT_POOL_MB")){ const long long mb = atoll(e); if(mb >= 64 && ((uint64_t)mb << 20) < c->pool_bytes) c->pool_bytes = (uint64_t)mb << 20; }
	}
#endif
	c->pool_bytes &= ~(uint64_t)4095; c->main_bytes = (c->pool_bytes / 2) & ~(uint64_t)4095;
	c->d_qid = c->d_cid = NULL; c->d_pairres = NULL; c->n_pairs = 0; c->d_alnres = NULL; c->n_items = 0; c->have_pairs = c->have_items = false;
	memset(&c->cnt, 0, sizeof c->cnt);
	c->cap_pairs = c->cap_items = 0;
#ifndef WTZ_EMUL
	c->arena.base = NULL; c->arena.cap = 0; c->arena.top = 0; c->arena.cache_bytes = 0;
	{ void *ab = NULL; const size_t acap = (size_t)3 << 29;      /* 1.5 GB */
	  if(hipMalloc(&ab, acap) == hipSuccess){ c->arena.base = (uint8_t*)ab; c->arena.cap = acap; } }
#endif
#ifndef WTZ_EMUL
	/* every failure from here on unwinds through wtz_ctx_destroy (all fields are initialised; it skips what does not exist yet) */
	if(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess
			|| hipStreamCreateWithFlags(&c->stream_mw, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_mw_fork, hipEventDisableTiming) != hipSuccess
			|| hipEventCreateWithFlags(&c->ev_mw_join, hipEventDisableTiming) != hipSuccess
			|| hipStreamCreateWithFlags(&c->stream_gap, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_gap_fork, hipEventDisableTiming) != hipSuccess
			|| hipEventCreateWithFlags(&c->ev_gap_join, hipEventDisableTiming) != hipSuccess){ wtz_ctx_destroy(c); return wtz_fail(WTZ_E_HIP, "hipStreamCreate / hipEventCreate failed"); }
	g_stream = c->stream;
	/* debugging switches are read once per context, not lazily from worker threads */
	c->env_sw_mode = 0; if(getenv("WTZ_SW_SCALAR") && atoi(getenv("WTZ_SW_SCALAR"))) c->env_sw_mode = 1; if(getenv("WTZ_SW_CHECK") && atoi(getenv("WTZ_SW_CHECK"))) c->env_sw_mode = 2;
	/* round 5: 0 = one wave per job always.  With the frame form at two waves per SIMD and the pool's counter sharded, the four-wave kernel (2.7x the SIMD time per row
	 * for 1.5x the speed of one job) only costs throughput: K-sw3 stage at configs[2] 729 ms with the long jobs (>= 512 rows) on four waves, 692 with >= 2048, 612 with none */
	c->env_use_reg = !(getenv("WTZ_SW_NOREG") && atoi(getenv("WTZ_SW_NOREG")));
	c->env_gap_side = (getenv("WTZ_GAP_SIDESTREAM") && atoi(getenv("WTZ_GAP_SIDESTREAM"))) ? 1 : 0;
	c->env_profile = getenv("WTZ_PROFILE_PAIR") != NULL;
	if(getenv("WTZ_DM_FIRST_BIG")) c->env_dm_first_big = atoi(getenv("WTZ_DM_FIRST_BIG"));
#endif
	c->env_trace = getenv("WTZ_STAGE_TRACE") != NULL;
	c->env_cand_stream = (getenv("WTZ_CAND_STREAM") && atoi(getenv("WTZ_CAND_STREAM")) != 0);
	if(getenv("WTZ_CAND_WG")) c->env_cand_wg = atoi(getenv("WTZ_CAND_WG"));
	if(getenv("WTZ_PAIR_HEAVY_FIRST")) c->env_heavy_first = atoi(getenv("WTZ_PAIR_HEAVY_FIRST"));
	if(getenv("WTZ_EXT_FR")) c->env_ext_fr = atoi(getenv("WTZ_EXT_FR"));
	if(getenv("WTZ_EXT_PK")) c->env_ext_pk = atoi(getenv("WTZ_EXT_PK"));
	if(getenv("WTZ_EXT_MW_ROWS")) c->env_ext_mw_rows = atoi(getenv("WTZ_EXT_MW_ROWS"));
	if(getenv("WTZ_EXT_FUSED")) c->env_ext_fused = atoi(getenv("WTZ_EXT_FUSED"));
	if(getenv("WTZ_ZREAD")) c->env_zread = atoi(getenv("WTZ_ZREAD"));
	if(getenv("WTZ_XCD_GROUP")) c->env_xcd_group = (uint32_t)atoi(getenv("WTZ_XCD_GROUP"));
	c->env_grp4 = (getenv("WTZ_WINALIGN4") && atoi(getenv("WTZ_WINALIGN4")) != 0);
	if(getenv("WTZ_WINALIGN_LANE")) c->env_lane = atoi(getenv("WTZ_WINALIGN_LANE"));
	if(getenv("WTZ_GAP_LANE")) c->env_gap_lane = atoi(getenv("WTZ_GAP_LANE"));
	c->env_fail_once = getenv("WTZ_POOL_FAIL_ONCE") != NULL;
	c->env_fail_at = getenv("WTZ_POOL_FAIL_AT") ? (unsigned)atoi(getenv("WTZ_POOL_FAIL_AT")) : 0u;
	c->env_tfail_at = getenv("WTZ_TPOOL_FAIL_AT") ? (unsigned)atoi(getenv("WTZ_TPOOL_FAIL_AT")) : 0u;
	int rc;
	if((rc = dev_alloc_persist((void**)&c->dP, sizeof(wtz_params_t))) || (rc = dev_h2d(c->dP, &c->P, sizeof(wtz_params_t))) ||
	   (rc = dev_alloc_persist((void**)&c->dpool, 2 * sizeof(wtz_pool_t)))){
		wtz_ctx_destroy(c); return rc;
	}
	/* a pool of the DEFAULT size is a share of what hipMemGetInfo reported a moment ago: another process on the device (parallel test workers, a second
	 * rank) may have taken it since.  Halve and try again down to 4 GB - the pool size never changes a result (pool exhaustion splits the batch) -;
	 * a size the caller asked for (--pool-gb) fails as it is. */
	for(;;){
#ifndef WTZ_EMUL
		/* a refused attempt that will be retried goes through the raw call: wtz_last_error must not keep the message of a failure that was recovered from */
		if(!pool_bytes && c->pool_bytes > (4ull << 30)){
			if(hipMalloc((void**)&c->pool_base, c->pool_bytes) == hipSuccess){ rc = WTZ_OK; break; }
			(void)hipGetLastError();              /* the refused allocation must not be what a later launch check reads */
			c->pool_base = NULL;
			uint64_t nb = c->pool_bytes / 2; if(nb < (4ull << 30)) nb = 4ull << 30;      /* never below the documented floor */
			c->pool_bytes = nb & ~(uint64_t)4095; c->main_bytes = (c->pool_bytes / 2) & ~(uint64_t)4095;
			continue;
		}
#endif
		rc = dev_alloc_persist((void**)&c->pool_base, c->pool_bytes);      /* the last attempt (or the caller's own size) fails loudly */
		break;
	}
	if(rc || (rc = pool_reset(c))){
		wtz_ctx_destroy(c); return rc;
	}
	*out = c;
	return WTZ_OK;
}

static void free_pending_index(wtz_ctx *c);
/* the table and the seed list of the previous build are kept for the next one of about the same size (a repeat of the step, the next shard of the same job) */
static void kpark(wtz_ctx *c, int w, void *p, size_t bytes){ if(!p) return; if(c->kpark_p[w]) dev_free_persist(c->kpark_p[w]); c->kpark_p[w] = p; c->kpark_b[w] = bytes; }
static void kflush(wtz_ctx *c){ for(int w = 0; w < 2; w++){ if(c->kpark_p[w]) dev_free_persist(c->kpark_p[w]); c->kpark_p[w] = NULL; c->kpark_b[w] = 0; } }
static int kalloc(wtz_ctx *c, int w, void **p, size_t n){
	if(c->kpark_p[w] && c->kpark_b[w] >= n && c->kpark_b[w] <= n + n / 8 + ((size_t)1 << 20)){ *p = c->kpark_p[w]; c->klive_b[w] = c->kpark_b[w]; c->kpark_p[w] = NULL; c->kpark_b[w] = 0; return WTZ_OK; }
	if(c->kpark_p[w]){ dev_free_persist(c->kpark_p[w]); c->kpark_p[w] = NULL; c->kpark_b[w] = 0; }
	int rc = dev_alloc_persist(p, n); if(rc == WTZ_OK) c->klive_b[w] = n; return rc;
}
static void free_kindex(wtz_ctx *c){
	if(!c->shares_indexes){
		if(c->klive_b[0]) kpark(c, 0, c->ktab, c->klive_b[0]); else dev_free_persist(c->ktab);
		if(c->klive_b[1]) kpark(c, 1, c->kseeds, c->klive_b[1]); else dev_free_persist(c->kseeds);
		c->klive_b[0] = c->klive_b[1] = 0;
	}
	c->ktab = NULL; c->kseeds = NULL; c->kmask = 0;
}
/* z-index allocation with recycling: a parked buffer of (nearly) the wanted size is taken instead of a fresh hipMalloc */
static int zalloc(wtz_ctx::zslot_t *z, void **p, size_t n){
	if(n == 0) n = 16;
	for(size_t i = 0; i < z->parked.size(); i++){
		if(z->parked[i].second >= n && z->parked[i].second <= n + n / 8 + 4096){
			*p = z->parked[i].first; z->live.push_back(z->parked[i]); z->parked.erase(z->parked.begin() + (long)i); return WTZ_OK;
		}
	}
	int rc = dev_alloc_persist(p, n); if(rc) return rc;
	z->live.push_back(std::make_pair(*p, n)); return WTZ_OK;
}
static void zpark_all(wtz_ctx::zslot_t *z){ for(size_t i = 0; i < z->live.size(); i++) z->parked.push_back(z->live[i]); z->live.clear(); }
static void zflush_parked(wtz_ctx::zslot_t *z){ for(size_t i = 0; i < z->parked.size(); i++) dev_free_persist(z->parked[i].first); z->parked.clear(); }
static void free_zindex(wtz_ctx *c){
	for(int k = 0; k < 2; k++){
		wtz_ctx::zslot_t *z = &c->zs[k];
		if(!c->shares_indexes){ zpark_all(z); zflush_parked(z); }      /* every z-index array comes from zalloc */
		z->zoff = NULL; memset(&z->Z, 0, sizeof z->Z); z->have = false; z->sub = false; z->sub_cap = 0; z->n_z = 0;
	}
}
static void free_batch(wtz_ctx *c){ c->n_pairs = 0; c->n_items = 0; c->have_pairs = false; c->have_items = false; }
static void free_batch_storage(wtz_ctx *c){
	dev_free_persist(c->d_qid); dev_free_persist(c->d_cid); dev_free_persist(c->d_pairres); dev_free_persist(c->d_alnres);
	c->d_qid = c->d_cid = NULL; c->d_pairres = NULL; c->d_alnres = NULL; c->cap_pairs = c->cap_items = 0; free_batch(c);
}
static int reserve_pairs(wtz_ctx *c, uint32_t n){
	if(n <= c->cap_pairs && c->d_pairres) return WTZ_OK;
	uint32_t cap = c->cap_pairs ? c->cap_pairs : 4096; while(cap < n) cap *= 2;
	(void)dev_sync();
	dev_free_persist(c->d_qid); dev_free_persist(c->d_cid); dev_free_persist(c->d_pairres); c->d_qid = c->d_cid = NULL; c->d_pairres = NULL; c->cap_pairs = 0;
	CHK(dev_alloc_persist((void**)&c->d_qid, (size_t)cap * 4)); CHK(dev_alloc_persist((void**)&c->d_cid, (size_t)cap * 4));
	CHK(dev_alloc_persist((void**)&c->d_pairres, (size_t)cap * sizeof(wtz_pairres_t)));
	c->cap_pairs = cap; return WTZ_OK;
}
static int reserve_items(wtz_ctx *c, uint32_t m){
	if(m <= c->cap_items && c->d_alnres) return WTZ_OK;
	uint32_t cap = c->cap_items ? c->cap_items : 4096; while(cap < m) cap *= 2;
	(void)dev_sync();
	dev_free_persist(c->d_alnres); c->d_alnres = NULL; c->cap_items = 0;
	CHK(dev_alloc_persist((void**)&c->d_alnres, (size_t)cap * sizeof(wtz_alnres_dev_t)));
	c->cap_items = cap; return WTZ_OK;
}

extern "C" void wtz_ctx_destroy(wtz_ctx_t *c){
	if(!c) return;
	{ CTX_ENTER(c); (void)dev_sync(); }
	free_batch_storage(c); free_kindex(c); if(!c->shares_indexes) kflush(c); free_zindex(c);
	dev_free_persist(c->d_text);
	dev_free_persist(c->cq_q); dev_free_persist(c->cq_nc); dev_free_persist(c->cq_cand); dev_free_persist(c->cq_bytes); dev_free_persist(c->cq_thr); dev_free_persist(c->cq_gptr);
	free_pending_index(c);
#ifndef WTZ_EMUL
	arena_cache_flush(&c->arena);
	if(c->arena.base) (void)hipFree(c->arena.base);
#endif
	if(!c->shares_indexes){ dev_free_persist(c->bits); dev_free_persist(c->rdoff); dev_free_persist(c->rdlen); }
	dev_free_persist(c->dP); dev_free_persist(c->dpool); dev_free_persist(c->pool_base);
#ifndef WTZ_EMUL
	if(c->stream_mw) (void)hipStreamDestroy(c->stream_mw);
	if(c->stream_gap) (void)hipStreamDestroy(c->stream_gap);
	if(c->stream_copy){ (void)hipStreamSynchronize(c->stream_copy); (void)hipStreamDestroy(c->stream_copy); if(c->ev_text_ready) (void)hipEventDestroy(c->ev_text_ready); for(int k = 0; k < 2; k++) if(c->ev_text_done[k]) (void)hipEventDestroy(c->ev_text_done[k]); }
	{ hipEvent_t evs[4] = { c->ev_mw_fork, c->ev_mw_join, c->ev_gap_fork, c->ev_gap_join }; for(int k = 0; k < 4; k++) if(evs[k]) (void)hipEventDestroy(evs[k]); }
	if(c->stream) (void)hipStreamDestroy(c->stream);
#endif
	delete c;
}

/* A second context on the same GPU that SHARES the parent's read-only device data (reads, k-mer table, z-index) and has its
 * own stream, scratch pool and per-batch state: lets a host thread keep another batch in flight.  The parent must outlive the
 * clone and must not rebuild its indexes while clones are in use (re-clone after wtz_index_build / wtz_zindex_build). */
extern "C" int wtz_ctx_clone(wtz_ctx_t *p, uint64_t pool_bytes, wtz_ctx_t **out){
	if(!p || !out) return wtz_fail(WTZ_E_ARG, "null argument");
	wtz_ctx_t *c = NULL;
	int rc = wtz_ctx_create(p->device, &p->P, pool_bytes ? pool_bytes : p->pool_bytes, &c);
	if(rc) return rc;
	c->shares_indexes = true;
	c->bits = p->bits; c->n_words = p->n_words; c->rdoff = p->rdoff; c->rdlen = p->rdlen; c->n_reads = p->n_reads; c->h_rdlen = p->h_rdlen;
	c->ktab = p->ktab; c->kmask = p->kmask; c->kseeds = p->kseeds; c->n_kocc = p->n_kocc;
	c->idx_beg = p->idx_beg; c->idx_end = p->idx_end; c->idx_len_sorted = p->idx_len_sorted;
	for(int k = 0; k < 2; k++){ c->zs[k].zoff = p->zs[k].zoff; c->zs[k].n_z = p->zs[k].n_z; c->zs[k].Z = p->zs[k].Z; c->zs[k].have = p->zs[k].have; }
	*out = c;
	return WTZ_OK;
}

extern "C" int wtz_upload_reads(wtz_ctx_t *c, const uint64_t *bits, uint64_t n_words, const uint64_t *rdoff, const uint32_t *rdlen, uint32_t n_reads){
	if(!c || !bits || !rdoff || !rdlen) return wtz_fail(WTZ_E_ARG, "null argument");
	CTX_ENTER(c);
	if(c->shares_indexes) return wtz_fail(WTZ_E_STATE, "wtz_upload_reads on a cloned context");
	dev_free_persist(c->bits); dev_free_persist(c->rdoff); dev_free_persist(c->rdlen); c->bits = NULL; c->rdoff = NULL; c->rdlen = NULL;
	free_kindex(c); free_zindex(c); free_batch(c);
	CHK(dev_alloc_persist((void**)&c->bits, (n_words + 2) * 8)); CHK(dev_set(c->bits, 0, (n_words + 2) * 8)); CHK(dev_h2d(c->bits, bits, n_words * 8));
	CHK(dev_alloc_persist((void**)&c->rdoff, (size_t)n_reads * 8)); CHK(dev_h2d(c->rdoff, rdoff, (size_t)n_reads * 8));
	CHK(dev_alloc_persist((void**)&c->rdlen, (size_t)n_reads * 4)); CHK(dev_h2d(c->rdlen, rdlen, (size_t)n_reads * 4));
	c->n_words = n_words; c->n_reads = n_reads; c->h_rdlen.assign(rdlen, rdlen + n_reads);
	return WTZ_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* f4: FASTA -> 2-bit on the device (seq2basebank, dna.h:397-410)                                    */
/* ------------------------------------------------------------------------------------------------ */
#include "wtz_ingest.h"
#ifndef WTZ_EMUL
/* dedicated streaming kernel: 256 threads, grid-stride over the HALF words of the chunk with four loads in flight per thread: a wave reads
 * 1 KB of text per instruction (16 bytes per lane, lanes contiguous) and writes 256 B of the bank (4 bytes per lane; the two halves of a
 * 64-bit word swap places: little-endian words, first base in the top bits) */
__global__ void __launch_bounds__(256) wtz_kernel_pack_ascii(const uint8_t *ascii, uint64_t n, uint64_t n_half, uint32_t *bits32, unsigned long long *n_pos, uint64_t *pos, uint64_t pos_cap, uint64_t pos_base){
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	for(; h + 3 * stride < n_half; h += 4 * stride){
		const uint32_t a = wtz_pack_half(ascii, n, h, n_pos, pos, pos_cap, pos_base), b = wtz_pack_half(ascii, n, h + stride, n_pos, pos, pos_cap, pos_base);
		const uint32_t c = wtz_pack_half(ascii, n, h + 2 * stride, n_pos, pos, pos_cap, pos_base), d = wtz_pack_half(ascii, n, h + 3 * stride, n_pos, pos, pos_cap, pos_base);
		bits32[h ^ 1] = a; bits32[(h + stride) ^ 1] = b; bits32[(h + 2 * stride) ^ 1] = c; bits32[(h + 3 * stride) ^ 1] = d;
	}
	for(; h < n_half; h += stride) bits32[h ^ 1] = wtz_pack_half(ascii, n, h, n_pos, pos, pos_cap, pos_base);
}
#endif
extern "C" int wtz_upload_reads_ascii(wtz_ctx_t *c, const char *seq, uint64_t n_bases, const uint64_t *rdoff, const uint32_t *rdlen, uint32_t n_reads, uint64_t rand_calls_before, uint64_t *n_random){
	if(!c || (!seq && n_bases) || !rdoff || !rdlen) return wtz_fail(WTZ_E_ARG, "null argument");
	CTX_ENTER(c);
	if(c->shares_indexes) return wtz_fail(WTZ_E_STATE, "wtz_upload_reads_ascii on a cloned context");
	dev_free_persist(c->bits); dev_free_persist(c->rdoff); dev_free_persist(c->rdlen); c->bits = NULL; c->rdoff = NULL; c->rdlen = NULL;
	free_kindex(c); free_zindex(c); free_batch(c);
	const uint64_t n_words = (n_bases + 31) / 32;
	c->n_words = 0; c->n_reads = 0; c->h_rdlen.clear();        /* "no reads uploaded" until the last chunk is packed: a failure below leaves the context in that state, not over a half-packed bank */
	const uint64_t CH = (uint64_t)256 << 20;             /* bases per chunk (a multiple of 32): 256 MB of text on the device at a time */
	uint8_t *d_txt = NULL; unsigned long long *d_np = NULL; uint64_t *d_pos = NULL; uint64_t pos_cap = (uint64_t)1 << 20;
	int rc = WTZ_OK;
	if((rc = dev_alloc_persist((void**)&c->bits, (n_words + 2) * 8)) || (rc = dev_set(c->bits, 0, (n_words + 2) * 8)) ||
	   (rc = dev_alloc_persist((void**)&c->rdoff, (size_t)n_reads * 8)) || (rc = dev_h2d(c->rdoff, rdoff, (size_t)n_reads * 8)) ||
	   (rc = dev_alloc_persist((void**)&c->rdlen, (size_t)n_reads * 4)) || (rc = dev_h2d(c->rdlen, rdlen, (size_t)n_reads * 4)) ||
	   (rc = dev_alloc_persist((void**)&d_txt, (size_t)WTZ_MIN(CH, n_bases) + 64)) || (rc = dev_alloc_persist((void**)&d_np, 16)) || (rc = dev_alloc_persist((void**)&d_pos, pos_cap * 8))){
		dev_free_persist(d_txt); dev_free_persist(d_np); dev_free_persist(d_pos);
		dev_free_persist(c->bits); dev_free_persist(c->rdoff); dev_free_persist(c->rdlen); c->bits = NULL; c->rdoff = NULL; c->rdlen = NULL;
		return rc;
	}
	uint64_t rank = rand_calls_before;
#ifndef WTZ_EMUL
	/* an empty launch first: the first kernel launch of a process loads the library's code object (several ms), which is not this kernel's time */
	hipLaunchKernelGGL(wtz_kernel_pack_ascii, dim3(1), dim3(256), 0, g_stream, (const uint8_t*)d_txt, (uint64_t)0, (uint64_t)0, (uint32_t*)c->bits, d_np, d_pos, pos_cap, (uint64_t)0);
	(void)hipStreamSynchronize(g_stream);
#endif
	for(uint64_t b0 = 0; b0 < n_bases && rc == WTZ_OK; b0 += CH){
		const uint64_t nb = WTZ_MIN(CH, n_bases - b0), nw = (nb + 31) / 32;
		if((rc = dev_h2d(d_txt, seq + b0, (size_t)nb))) break;
		for(;;){
			if((rc = dev_set(d_np, 0, 16))) break;
			uint64_t *bits = c->bits + b0 / 32; unsigned long long *np = d_np; uint64_t *pos = d_pos; const uint8_t *txt = d_txt; const uint64_t cap = pos_cap;
			wtz_timer tm; tm.start();
#ifndef WTZ_EMUL
			{ const uint64_t nh = nw * 2; uint64_t nblk = (nh + 1023) / 1024; if(nblk > 256 * 32) nblk = 256 * 32; if(nblk < 1) nblk = 1;      /* at most 32 workgroups per CU; stride is even, so h ^ 1 stays inside the chunk's words */
			  hipLaunchKernelGGL(wtz_kernel_pack_ascii, dim3((uint32_t)nblk), dim3(256), 0, g_stream, txt, nb, nh, (uint32_t*)bits, np, pos, cap, b0);
			  if(hipGetLastError() != hipSuccess){ rc = wtz_fail(WTZ_E_HIP, "wtz_kernel_pack_ascii launch failed"); break; } }
#else
			for(uint64_t h = 0; h < nw * 2; h++) ((uint32_t*)bits)[h ^ 1] = wtz_pack_half(txt, nb, h, np, pos, cap, b0);
#endif
			c->cnt.ms_ingest += tm.stop();                          /* HIP events around the kernel alone */
			unsigned long long cnt = 0;
			if((rc = dev_d2h(&cnt, d_np, 8))) break;
			if(cnt > pos_cap){       /* more non-bases than the list holds: grow it and pack the chunk again */
				dev_free_persist(d_pos); d_pos = NULL; pos_cap = cnt + cnt / 4;
				if((rc = dev_alloc_persist((void**)&d_pos, pos_cap * 8))) break;
				continue;
			}
			if(cnt){
				std::vector<uint64_t> hp((size_t)cnt);
				if((rc = dev_d2h(hp.data(), d_pos, (size_t)cnt * 8))) break;
				std::sort(hp.begin(), hp.end());                    /* file order = ascending position */
				if((rc = dev_h2d(d_pos, hp.data(), (size_t)cnt * 8))) break;
				uint64_t *allbits = c->bits; const uint64_t r0 = rank;
				wtz_timer tf; tf.start();
				if((rc = wtz_launch<K_pack_fix>(0, cnt, [=] WTZ_LAMBDA (uint64_t r){ wtz_fix_random_base(r, pos, r0, allbits); }))) break;
				if((rc = dev_sync())) break;
				c->cnt.ms_ingest += tf.stop();
				rank += cnt;
			}
			break;
		}
	}
	dev_free_persist(d_txt); dev_free_persist(d_np); dev_free_persist(d_pos);
	if(rc != WTZ_OK){ dev_free_persist(c->bits); dev_free_persist(c->rdoff); dev_free_persist(c->rdlen); c->bits = NULL; c->rdoff = NULL; c->rdlen = NULL; return rc; }
	c->n_words = n_words; c->n_reads = n_reads; c->h_rdlen.assign(rdlen, rdlen + n_reads);
	c->cnt.bytes_ingest_algo += n_bases + n_words * 8;
	if(n_random) *n_random = rank - rand_calls_before;
	return WTZ_OK;
}
extern "C" int wtz_append_revcomp_views(wtz_ctx_t *c){
	if(!c || !c->bits) return wtz_fail(WTZ_E_ARG, "reads not uploaded");
	CTX_ENTER(c);
	if(c->shares_indexes) return wtz_fail(WTZ_E_STATE, "wtz_append_revcomp_views on a cloned context");
	free_kindex(c); free_zindex(c); free_batch(c);
	const uint32_t n = c->n_reads;
	if((uint64_t)n * 2 > 0xFFFFFFFFull) return wtz_fail(WTZ_E_ARG, "too many reads for their reverse-complement views");
	std::vector<uint64_t> h_off((size_t)n * 2), vw((size_t)n + 1);      /* vw[i] = first word of view i behind the old bank */
	CHK(dev_d2h(h_off.data(), c->rdoff, (size_t)n * 8));
	uint64_t words = 0;
	for(uint32_t i = 0; i < n; i++){ vw[i] = words; words += ((uint64_t)c->h_rdlen[i] + 31) / 32; }
	vw[n] = words;
	const uint64_t old_w = c->n_words, new_w = old_w + words;
	uint64_t *nb = NULL; uint64_t *nro = NULL; uint32_t *nrl = NULL;
	CHK(dev_alloc_persist((void**)&nb, (new_w + 2) * 8)); CHK(dev_set(nb + old_w, 0, (words + 2) * 8)); CHK(dev_d2d(nb, c->bits, old_w * 8));
	std::vector<uint32_t> h_len((size_t)n * 2);
	for(uint32_t i = 0; i < n; i++){ h_len[i] = c->h_rdlen[i]; h_len[n + i] = c->h_rdlen[i]; h_off[n + i] = (old_w + vw[i]) * 32; }
	CHK(dev_alloc_persist((void**)&nro, (size_t)n * 2 * 8)); CHK(dev_h2d(nro, h_off.data(), (size_t)n * 2 * 8));
	CHK(dev_alloc_persist((void**)&nrl, (size_t)n * 2 * 4)); CHK(dev_h2d(nrl, h_len.data(), (size_t)n * 2 * 4));
	uint64_t *d_vw = NULL; CHK(dev_alloc((void**)&d_vw, ((size_t)n + 1) * 8)); CHK(dev_h2d(d_vw, vw.data(), ((size_t)n + 1) * 8));
	const uint64_t *src = c->bits; const uint64_t *ro = nro; const uint32_t *rl = nrl; uint64_t *dst = nb + old_w;
	CHK(wtz_launch<K_revcomp_views>(0, words, [=] WTZ_LAMBDA (uint64_t w){
		uint32_t lo = 0, hi = n;                         /* the view that holds word w: last i with vw[i] <= w */
		while(hi - lo > 1){ const uint32_t mid = (lo + hi) >> 1; if(d_vw[mid] <= w) lo = mid; else hi = mid; }
		dst[w] = wtz_revcomp_word(src, ro[lo], rl[lo], (uint32_t)(w - d_vw[lo]));
	}));
	CHK(dev_sync());
	dev_free_persist(c->bits); dev_free_persist(c->rdoff); dev_free_persist(c->rdlen);
	c->bits = nb; c->rdoff = nro; c->rdlen = nrl; c->n_words = new_w; c->n_reads = n * 2; c->h_rdlen = h_len;
	return WTZ_OK;
}
extern "C" int wtz_fetch_read_bits(wtz_ctx_t *c, uint64_t *bits, uint64_t n_words){
	if(!c || !bits || !c->bits) return wtz_fail(WTZ_E_ARG, "reads not uploaded / null argument");
	if(n_words > c->n_words) return wtz_fail(WTZ_E_ARG, "wtz_fetch_read_bits: %llu words asked, %llu uploaded", (unsigned long long)n_words, (unsigned long long)c->n_words);
	CTX_ENTER(c);
	CHK(dev_d2h(bits, c->bits, (size_t)n_words * 8));
	return WTZ_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* A2: k-mer index                                                                                   */
/* ------------------------------------------------------------------------------------------------ */
extern "C" int wtz_index_build(wtz_ctx_t *c, uint32_t id_beg, uint32_t id_end, uint32_t *max_kmer_freq, wtz_index_stats_t *stats){
	if(!c || !c->bits || !max_kmer_freq) return wtz_fail(WTZ_E_ARG, "reads not uploaded / null argument");
	if(id_end > c->n_reads) id_end = c->n_reads;
	if(id_beg > id_end) id_beg = id_end;
	const uint32_t nr = id_end - id_beg;
	CTX_ENTER(c);
	if(c->shares_indexes) return wtz_fail(WTZ_E_STATE, "wtz_index_build on a cloned context");
	const bool prof_ix = c->env_profile; double tix[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tix0 = wtz_wall();
	auto lapix = [&](int k){ if(prof_ix){ (void)dev_sync(); const double t = wtz_wall(); tix[k] += t - tix0; tix0 = t; } };
	free_kindex(c);
	lapix(0);
	wtz_timer tm; tm.start();
	const wtz_reads_t R = ctx_reads(c); const uint32_t ksize = c->P.ksize, hk = c->P.hk, ksave = c->P.ksave;
	/* the walk of a read is a serial recurrence, but it restarts exactly anywhere (wtz_walk_warm_start): one lane per
	 * WTZ_WALK_CHUNK-base piece instead of one per read, pieces listed in read order */
	std::vector<uint32_t> p_rid, p_jb;
	for(uint32_t r = id_beg; r < id_end; r++) for(uint32_t jb = 0; jb == 0 || jb < c->h_rdlen[r]; jb += WTZ_WALK_CHUNK){ p_rid.push_back(r); p_jb.push_back(jb); }
	const size_t np = p_rid.size();
	uint32_t *d_prid = NULL, *d_pjb = NULL;
	CHK(dev_alloc((void**)&d_prid, (np + 1) * 4)); CHK(dev_alloc((void**)&d_pjb, (np + 1) * 4));
	CHK(dev_h2d(d_prid, p_rid.data(), np * 4)); CHK(dev_h2d(d_pjb, p_jb.data(), np * 4));
	lapix(1);
	uint64_t *d_cnt = NULL; CHK(dev_alloc((void**)&d_cnt, (np + 1) * 8));
	CHK(wtz_launch<K_kcount>(0, np, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_kcount((uint32_t)t, R, d_prid, d_pjb, ksize, hk, ksave, d_cnt); }));
	std::vector<uint64_t> h_cnt(np + 1);
	CHK(dev_d2h(h_cnt.data(), d_cnt, np * 8));
	uint64_t tot = 0; for(size_t i = 0; i < np; i++){ uint64_t v = h_cnt[i]; h_cnt[i] = tot; tot += v; } h_cnt[np] = tot;
	CHK(dev_h2d(d_cnt, h_cnt.data(), (np + 1) * 8));
	lapix(2);
	uint64_t *d_keys = NULL; uint32_t *d_vals = NULL;
	CHK(dev_alloc((void**)&d_keys, (tot + 1) * 8)); CHK(kalloc(c, 1, (void**)&d_vals, (tot + 1) * 4));
	lapix(3);
	CHK(wtz_launch<K_kfill>(0, np, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_kfill((uint32_t)t, R, d_prid, d_pjb, ksize, hk, ksave, d_cnt, d_keys, d_vals); }));
	CHK(dev_sync());
	dev_free(d_cnt); dev_free(d_prid); dev_free(d_pjb);
	(void)nr;
	lapix(4);
	CHK(dev_sort_pairs_u64_u32(d_keys, d_vals, tot, 2 * ksize));
	lapix(5);
	unsigned long long *d_stat = NULL; CHK(dev_alloc((void**)&d_stat, 4 * 8)); CHK(dev_set(d_stat, 0, 4 * 8));
	const uint64_t n_str = tot < (1ull << 18) ? (tot ? tot : 1) : (1ull << 18);      /* strided counting passes: one atomic per wavefront at the end */
	CHK(wtz_launch<K_kstats>(0, n_str, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_kstats_stride(t, n_str, d_keys, tot, d_stat + 0, d_stat + 1); }));
	unsigned long long h_stat[4]; CHK(dev_d2h(h_stat, d_stat, 4 * 8));
	const uint64_t ktot = tot - h_stat[0], ktyp = h_stat[1];     /* d_stat[0] accumulates the saturation excess */
	uint32_t K = *max_kmer_freq;
	if(K < 2){ uint32_t kavg = (uint32_t)(ktot / (ktyp + 1)); if(kavg < 20) kavg = 20; K = kavg * 5; }       /* wtzmo.c:380-393 */
	*max_kmer_freq = K;
	CHK(wtz_launch<K_kinsert>(0, n_str, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_kkept_stride(t, n_str, d_keys, tot, K, d_stat + 2); }));
	CHK(dev_d2h(h_stat, d_stat, 4 * 8));
	const uint64_t n_kept = h_stat[2];
	uint64_t cap = 1024; while(cap < n_kept * 2 + 2) cap <<= 1;
	CHK(kalloc(c, 0, (void**)&c->ktab, cap * sizeof(wtz_kslot_t))); CHK(dev_set(c->ktab, 0xFF, cap * sizeof(wtz_kslot_t)));
	c->kmask = cap - 1;
	wtz_kslot_t *tab = c->ktab; const uint64_t kmask = c->kmask;
	CHK(wtz_launch<K_kinsert>(0, tot, [=] WTZ_LAMBDA (uint64_t i){ wtz_task_kinsert(i, d_keys, tot, K, tab, kmask, d_stat + 2); }));
	CHK(dev_sync());
	dev_free(d_keys); dev_free(d_stat);
	c->kseeds = d_vals; c->n_kocc = tot;
	c->idx_beg = id_beg; c->idx_end = id_end; c->idx_len_sorted = true;
	for(uint32_t r = id_beg; r + 1 < id_end; r++) if(c->h_rdlen[r] < c->h_rdlen[r + 1]){ c->idx_len_sorted = false; break; }
	c->cnt.ms_index += tm.stop();
	lapix(6);
	if(prof_ix) fprintf(stderr, "[index-profile] ms: free %.1f pieces+h2d %.1f count %.1f alloc %.1f fill %.1f sort %.1f table %.1f\n", tix[0] * 1e3, tix[1] * 1e3, tix[2] * 1e3, tix[3] * 1e3, tix[4] * 1e3, tix[5] * 1e3, tix[6] * 1e3);
	if(stats){
		stats->n_occ = tot; stats->n_distinct = ktyp; stats->ktot = ktot; stats->n_kept = n_kept; stats->max_kmer_freq = K;
		uint64_t tl = 0; for(uint32_t i = 0; i < c->n_reads; i++) tl += c->h_rdlen[i];
		stats->avg_rdlen = c->n_reads ? (uint32_t)(tl / c->n_reads) : 10000;
	}
	return WTZ_OK;
}


/* ------------------------------------------------------------------------------------------------ */
/* A2 sharded by read-id range (see include/wtzmo_hip.h)                                             */
/* ------------------------------------------------------------------------------------------------ */
static void free_pending_index(wtz_ctx *c){
	dev_free_persist(c->pend_keys); dev_free_persist(c->pend_vals); dev_free_persist(c->pend_dk); dev_free_persist(c->pend_dc); dev_free_persist(c->pend_dstart);
	c->pend_keys = NULL; c->pend_vals = NULL; c->pend_dk = NULL; c->pend_dc = NULL; c->pend_dstart = NULL; c->pend_tot = 0; c->pend_nd = 0;
}
extern "C" int wtz_index_count(wtz_ctx_t *c, uint32_t id_beg, uint32_t id_end, uint64_t *n_distinct, uint64_t *n_occ){
	if(!c || !c->bits || !n_distinct) return wtz_fail(WTZ_E_ARG, "reads not uploaded / null argument");
	if(id_end > c->n_reads) id_end = c->n_reads;
	if(id_beg > id_end) id_beg = id_end;
	CTX_ENTER(c);
	if(c->shares_indexes) return wtz_fail(WTZ_E_STATE, "wtz_index_count on a cloned context");
	free_kindex(c); free_pending_index(c);
	wtz_timer tm; tm.start();
	const wtz_reads_t R = ctx_reads(c); const uint32_t ksize = c->P.ksize, hk = c->P.hk, ksave = c->P.ksave;
	std::vector<uint32_t> p_rid, p_jb;
	for(uint32_t r = id_beg; r < id_end; r++) for(uint32_t jb = 0; jb == 0 || jb < c->h_rdlen[r]; jb += WTZ_WALK_CHUNK){ p_rid.push_back(r); p_jb.push_back(jb); }
	const size_t np = p_rid.size();
	uint32_t *d_prid = NULL, *d_pjb = NULL; uint64_t *d_cnt = NULL;
	CHK(dev_alloc((void**)&d_prid, (np + 1) * 4)); CHK(dev_alloc((void**)&d_pjb, (np + 1) * 4)); CHK(dev_alloc((void**)&d_cnt, (np + 1) * 8));
	CHK(dev_h2d(d_prid, p_rid.data(), np * 4)); CHK(dev_h2d(d_pjb, p_jb.data(), np * 4));
	CHK(wtz_launch<K_kcount>(0, np, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_kcount((uint32_t)t, R, d_prid, d_pjb, ksize, hk, ksave, d_cnt); }));
	std::vector<uint64_t> h_cnt(np + 1);
	CHK(dev_d2h(h_cnt.data(), d_cnt, np * 8));
	uint64_t tot = 0; for(size_t i = 0; i < np; i++){ uint64_t v = h_cnt[i]; h_cnt[i] = tot; tot += v; } h_cnt[np] = tot;
	CHK(dev_h2d(d_cnt, h_cnt.data(), (np + 1) * 8));
	if(tot >= 0xFFFFFFFFull) return wtz_fail(WTZ_E_ARG, "wtz_index_count: more than 2^32 k-mer occurrences in one shard; use more shards");
	CHK(dev_alloc_persist((void**)&c->pend_keys, (tot + 1) * 8)); CHK(dev_alloc_persist((void**)&c->pend_vals, (tot + 1) * 4));
	uint64_t *d_keys = c->pend_keys; uint32_t *d_vals = c->pend_vals;
	CHK(wtz_launch<K_kfill>(0, np, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_kfill((uint32_t)t, R, d_prid, d_pjb, ksize, hk, ksave, d_cnt, d_keys, d_vals); }));
	CHK(dev_sync());
	CHK(dev_sort_pairs_u64_u32(d_keys, d_vals, tot, 2 * ksize));
	uint32_t *d_flag = NULL, *d_dpos = NULL;
	CHK(dev_alloc((void**)&d_flag, (tot + 2) * 4)); CHK(dev_alloc((void**)&d_dpos, (tot + 2) * 4)); CHK(dev_set(d_flag, 0, (tot + 2) * 4));
	CHK(wtz_launch<K_khead>(0, tot, [=] WTZ_LAMBDA (uint64_t i){ wtz_task_khead(i, d_keys, d_flag); }));
	CHK(dev_exclusive_scan_u32(d_flag, d_dpos, tot + 1));
	uint32_t nd = 0; CHK(dev_d2h(&nd, d_dpos + tot, 4));
	CHK(dev_alloc_persist((void**)&c->pend_dk, ((size_t)nd + 1) * 8)); CHK(dev_alloc_persist((void**)&c->pend_dc, ((size_t)nd + 1) * 4)); CHK(dev_alloc_persist((void**)&c->pend_dstart, ((size_t)nd + 1) * 8));
	uint64_t *dk = c->pend_dk, *dst = c->pend_dstart; uint32_t *dc = c->pend_dc;
	CHK(wtz_launch<K_kdistinct>(0, tot, [=] WTZ_LAMBDA (uint64_t i){ wtz_task_kdistinct(i, d_keys, tot, d_flag, d_dpos, dk, dc, dst); }));
	CHK(dev_sync());
	c->pend_tot = tot; c->pend_nd = nd; c->pend_beg = id_beg; c->pend_end = id_end;
	c->cnt.ms_index += tm.stop();
	*n_distinct = nd; if(n_occ) *n_occ = tot;
	return WTZ_OK;
}
extern "C" int wtz_index_counts_fetch(wtz_ctx_t *c, uint64_t *kmers, uint32_t *cnts){
	if(!c || !kmers || !cnts) return wtz_fail(WTZ_E_ARG, "null argument");
	if(!c->pend_keys) return wtz_fail(WTZ_E_STATE, "wtz_index_counts_fetch before wtz_index_count");
	CTX_ENTER(c);
	CHK(dev_d2h(kmers, c->pend_dk, (size_t)c->pend_nd * 8)); CHK(dev_d2h(cnts, c->pend_dc, (size_t)c->pend_nd * 4));
	return WTZ_OK;
}
extern "C" int wtz_index_finish(wtz_ctx_t *c, const uint32_t *total_cnt, uint32_t K, uint64_t *n_kept_out){
	if(!c || (!total_cnt && c && c->pend_nd)) return wtz_fail(WTZ_E_ARG, "null argument");
	if(!c->pend_keys) return wtz_fail(WTZ_E_STATE, "wtz_index_finish before wtz_index_count");
	CTX_ENTER(c);
	wtz_timer tm; tm.start();
	const uint64_t nd = c->pend_nd;
	uint32_t *d_tc = NULL; unsigned long long *d_stat = NULL;
	CHK(dev_alloc((void**)&d_tc, (nd + 1) * 4)); CHK(dev_h2d(d_tc, total_cnt, nd * 4));
	CHK(dev_alloc((void**)&d_stat, 8)); CHK(dev_set(d_stat, 0, 8));
	const uint64_t *dk = c->pend_dk, *dst = c->pend_dstart; const uint32_t *dc = c->pend_dc;
	CHK(wtz_launch<K_kinsert_total>(0, nd, [=] WTZ_LAMBDA (uint64_t d){ wtz_task_kinsert_total(d, dk, dc, dst, d_tc, K, (wtz_kslot_t*)NULL, 0, d_stat); }));
	unsigned long long n_kept = 0; CHK(dev_d2h(&n_kept, d_stat, 8));
	uint64_t cap = 1024; while(cap < n_kept * 2 + 2) cap <<= 1;
	CHK(dev_alloc_persist((void**)&c->ktab, cap * sizeof(wtz_kslot_t))); CHK(dev_set(c->ktab, 0xFF, cap * sizeof(wtz_kslot_t)));
	c->kmask = cap - 1;
	wtz_kslot_t *tab = c->ktab; const uint64_t kmask = c->kmask;
	CHK(wtz_launch<K_kinsert_total>(0, nd, [=] WTZ_LAMBDA (uint64_t d){ wtz_task_kinsert_total(d, dk, dc, dst, d_tc, K, tab, kmask, d_stat); }));
	CHK(dev_sync());
	c->kseeds = c->pend_vals; c->pend_vals = NULL; c->n_kocc = c->pend_tot;
	c->idx_beg = c->pend_beg; c->idx_end = c->pend_end; c->idx_len_sorted = true;
	for(uint32_t r = c->idx_beg; r + 1 < c->idx_end; r++) if(c->h_rdlen[r] < c->h_rdlen[r + 1]){ c->idx_len_sorted = false; break; }
	free_pending_index(c);
	c->cnt.ms_index += tm.stop();
	if(n_kept_out) *n_kept_out = n_kept;
	return WTZ_OK;
}
extern "C" void wtz_cand_tail_host(const uint64_t *groups, uint32_t ng, uint32_t kovl, uint32_t ncand, uint64_t *heap, uint32_t *hn){ wtz_cand_tail(groups, ng, kovl, ncand, heap, hn); }

/* ------------------------------------------------------------------------------------------------ */
/* A5: z-mer index of every read                                                                     */
/* ------------------------------------------------------------------------------------------------ */
/* members == NULL: the z-index of every read.  Else (ascending read ids): of those reads only - every other read gets an empty slice, so the
 * kernels address the index exactly as before.  The subset form is rebuilt per batch of queries (their candidate sets bound what a batch
 * can look up), which is what lets a 10 Gbp read set (160 GB of z-index at 16 B per base) run in 288 GB: its arrays are allocated once
 * with head-room and reused. */
static int zindex_build_impl(wtz_ctx_t *c, const uint32_t *members, uint32_t nm, int slot = 0){
	if(!c || !c->bits) return wtz_fail(WTZ_E_ARG, "reads not uploaded");
	CTX_ENTER(c);
	if(c->shares_indexes) return wtz_fail(WTZ_E_STATE, "wtz_zindex_build on a cloned context");
	const bool subset = members != NULL;
	wtz_ctx::zslot_t *z = &c->zs[slot];
	if(!subset || !z->sub){ zpark_all(z); if(subset) zflush_parked(z); z->zoff = NULL; memset(&z->Z, 0, sizeof z->Z); z->sub_cap = 0; }      /* the old arrays are recycled below */
	z->have = false;
	if(slot == 0) c->zs[1].have = false;       /* a query-side index belongs to the batch it was built for */
	wtz_timer tm; tm.start();
	const wtz_reads_t R = ctx_reads(c); const uint32_t nr = c->n_reads, zsize = c->P.zsize, hz = c->P.hz, zcut = c->P.max_zmer_freq;
	if(z->zoff == NULL) CHK(zalloc(z, (void**)&z->zoff, ((size_t)nr + 1) * 8));
	uint64_t *d_off = z->zoff;
	std::vector<uint32_t> p_rid, p_jb; std::vector<size_t> first_piece((size_t)nr + 1);
	{ uint32_t mi = 0;
	  for(uint32_t r = 0; r < nr; r++){
		first_piece[r] = p_rid.size();
		if(subset){ if(mi < nm && members[mi] == r) mi++; else continue; }
		for(uint32_t jb = 0; jb == 0 || jb < c->h_rdlen[r]; jb += WTZ_WALK_CHUNK){ p_rid.push_back(r); p_jb.push_back(jb); }
	  }
	  if(subset && mi != nm) return wtz_fail(WTZ_E_ARG, "wtz_zindex_build_subset: the read ids must be ascending, unique and in range");
	}
	const size_t np = p_rid.size(); first_piece[nr] = np;
	uint32_t *d_prid = NULL, *d_pjb = NULL; uint64_t *d_poff = NULL;
	CHK(dev_alloc((void**)&d_prid, (np + 1) * 4)); CHK(dev_alloc((void**)&d_pjb, (np + 1) * 4)); CHK(dev_alloc((void**)&d_poff, (np + 1) * 8));
	CHK(dev_h2d(d_prid, p_rid.data(), np * 4)); CHK(dev_h2d(d_pjb, p_jb.data(), np * 4));
	CHK(wtz_launch<K_zcount>(0, np, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_zcount((uint32_t)t, R, d_prid, d_pjb, zsize, hz, d_poff); }));
	std::vector<uint64_t> hp(np + 1), h((size_t)nr + 1);
	CHK(dev_d2h(hp.data(), d_poff, np * 8));
	uint64_t tot = 0; for(size_t i = 0; i < np; i++){ uint64_t v = hp[i]; hp[i] = tot; tot += v; } hp[np] = tot;
	for(uint32_t r = 0; r <= nr; r++) h[r] = hp[first_piece[r]];
	CHK(dev_h2d(d_poff, hp.data(), (np + 1) * 8));
	CHK(dev_h2d(d_off, h.data(), ((size_t)nr + 1) * 8));
	z->n_z = tot;
	wtz_zindex_t Z; memset(&Z, 0, sizeof Z); Z.zoff = z->zoff;
	if(subset && z->sub && tot + 1 <= z->sub_cap) Z = z->Z;      /* the arrays of the previous subset are large enough */
	else {
		if(subset && z->sub){ (void)dev_sync(); zpark_all(z); zflush_parked(z); CHK(zalloc(z, (void**)&z->zoff, ((size_t)nr + 1) * 8)); d_off = z->zoff; CHK(dev_h2d(d_off, h.data(), ((size_t)nr + 1) * 8)); Z.zoff = z->zoff; }
		const uint64_t cap = subset ? tot + tot / 4 + 1024 : tot + 1;      /* subsets: head-room, so that most batches reuse the allocation */
		CHK(zalloc(z, (void**)&Z.mer, cap * 4)); CHK(zalloc(z, (void**)&Z.pos, cap * 4)); CHK(zalloc(z, (void**)&Z.len, cap * 2));
		CHK(zalloc(z, (void**)&Z.ok, cap)); CHK(zalloc(z, (void**)&Z.sidx, cap * 4));
		CHK(zalloc(z, (void**)&Z.dmer, cap * 4)); CHK(zalloc(z, (void**)&Z.dfirst, cap * 4)); CHK(zalloc(z, (void**)&Z.dcnt, cap * 2));
		CHK(zalloc(z, (void**)&Z.dn, ((size_t)nr + 1) * 4));
		zflush_parked(z);                     /* whatever did not fit a request goes back to the driver */
		z->sub_cap = subset ? cap : 0;
	}
	z->sub = subset;
	z->Z = Z;
	{
		/* chunks of consecutive reads, so that the temporaries (sort keys and their double buffer, run flags / lengths / ranks: 32 B per z-mer beside the 25 B the
		 * index keeps) are bounded by the chunk and not by the read set: every step below is per read.  WTZ_ZCHUNK_M: z-mers per chunk in millions */
		static uint64_t chunk_z = 0;
		if(!chunk_z){ const char *e = getenv("WTZ_ZCHUNK_M"); chunk_z = (uint64_t)((e && atof(e) > 0 ? atof(e) : 256.0) * 1e6); if(chunk_z < 1) chunk_z = 1; }
		unsigned rbits = 1; while((1ull << rbits) < (uint64_t)nr + 1) rbits++;
		/* reads whose z-mers fit the LDS of a CU are indexed by one workgroup each (wtz_task_zread); the ids are in length order, so what does not fit is a
		 * prefix [0, rL) of the ids (plus whatever short read sits among them): that prefix goes through the device-wide form in chunks */
		uint32_t rL = nr;
		if(c->env_zread){
			rL = 0;
			static const uint32_t cls[7] = { 2048u, 3072u, 4096u, 6144u, 8192u, 12288u, 16384u };      /* LDS per workgroup follows the class: finer classes = more workgroups per CU */
			for(uint32_t r = 0; r < nr; r++) if(h[r + 1] - h[r] > WTZ_ZR_MAXN || c->h_rdlen[r] > WTZ_ZR_MAXLEN(WTZ_ZR_MAXN)) rL = r + 1;
			std::vector<uint32_t> lst[7];
			for(uint32_t r = rL; r < nr; r++){
				const uint64_t nz = h[r + 1] - h[r]; if(!nz) continue;
				int k = 0; while(k < 6 && (nz > cls[k] || c->h_rdlen[r] > WTZ_ZR_MAXLEN(cls[k]))) k++;      /* the class holds the read's z-mers and its bases */
				lst[k].push_back(r);
			}
			if(nr > rL){ uint32_t *dn = Z.dn + rL; CHK(dev_set(dn, 0, (size_t)(nr - rL) * 4)); }
			for(int k = 6; k >= 0; k--){
				if(lst[k].empty()) continue;
				const uint32_t np = cls[k], nth = 512u, ldsb = wtz_zr_lds_bytes(np);
				uint32_t *d_lst = NULL; CHK(dev_alloc((void**)&d_lst, lst[k].size() * 4)); CHK(dev_h2d(d_lst, lst[k].data(), lst[k].size() * 4));
#ifdef WTZ_EMUL
				std::vector<uint32_t> emul_lds(ldsb / 4 + 16); uint32_t *lds_emul = emul_lds.data();
				CHK(wtz_launch_wg<K_zread>(lst[k].size(), [=] WTZ_LAMBDA (uint64_t t){ wtz_task_zread(d_lst[t], R, zsize, hz, zcut, Z, lds_emul, np); }, 1u, 0u));
#else
				CHK(wtz_launch_wg<K_zread>(lst[k].size(), [=] WTZ_LAMBDA (uint64_t t){ wtz_task_zread(d_lst[t], R, zsize, hz, zcut, Z, (uint32_t*)wtz_wave_scratch(), np); }, nth, ldsb));
#endif
			}
			CHK(dev_sync());
		}
		uint32_t r0 = 0;
		while(r0 < rL){
			uint32_t r1 = r0 + 1;
			while(r1 < rL && h[r1 + 1] - h[r0] <= chunk_z) r1++;
			const uint64_t base = h[r0], n = h[r1] - h[r0];
			const size_t p0 = first_piece[r0], p1 = first_piece[r1];
			if(n){
#ifndef WTZ_EMUL
				wtz_arena_scope chunk_scope(g_arena);      /* dev_free is a no-op inside an API call: the chunk's temporaries go back (to the arena / its cache) when this scope ends */
#endif
				uint64_t *d_key = NULL; uint32_t *d_flag = NULL, *d_cnt = NULL, *d_dpos = NULL; uint32_t *d_val = Z.sidx;
				CHK(dev_alloc((void**)&d_key, (n + 1) * 8));
				CHK(wtz_launch<K_zfill>(0, p1 - p0, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_zfill((uint32_t)(p0 + t), R, d_prid, d_pjb, d_poff, zsize, hz, Z, d_key, d_val, base); }));
				CHK(dev_sort_pairs_u64_u32(d_key, d_val + base, n, 32 + rbits));          /* stable: positions ascend inside a (read, mer) run */
				CHK(dev_alloc((void**)&d_flag, (n + 2) * 4)); CHK(dev_alloc((void**)&d_cnt, (n + 2) * 4)); CHK(dev_alloc((void**)&d_dpos, (n + 2) * 4));
				CHK(dev_set(d_flag, 0, (n + 2) * 4));
				const uint32_t *d_valb = d_val + base;
				CHK(wtz_launch<K_zrun>(0, n, [=] WTZ_LAMBDA (uint64_t i){ wtz_task_zrun(i, d_key, d_valb, n, zcut, Z, d_flag, d_cnt); }));
				CHK(dev_exclusive_scan_u32(d_flag, d_dpos, n + 1));
				CHK(wtz_launch<K_zdistinct>(0, n, [=] WTZ_LAMBDA (uint64_t i){ wtz_task_zdistinct(i, d_key, d_flag, d_cnt, d_dpos, Z, base); }));
				CHK(wtz_launch<K_zdn>(0, r1 - r0, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_zdn(r0 + (uint32_t)t, d_dpos, Z, base); }));
				CHK(dev_sync());
				dev_free(d_key); dev_free(d_flag); dev_free(d_cnt); dev_free(d_dpos);
			} else {
				uint32_t *dn = Z.dn + r0; CHK(dev_set(dn, 0, (size_t)(r1 - r0) * 4));
			}
			r0 = r1;
		}
	}
	dev_free(d_prid); dev_free(d_pjb); dev_free(d_poff);
	z->have = true;
	c->cnt.ms_zindex += tm.stop();
	return WTZ_OK;
}
extern "C" int wtz_zindex_build(wtz_ctx_t *c){ return zindex_build_impl(c, NULL, 0); }
extern "C" int wtz_zindex_build_subset(wtz_ctx_t *c, const uint32_t *ids, uint32_t n){
	if(!ids && n) return wtz_fail(WTZ_E_ARG, "null argument");
	static const uint32_t none = 0;
	return zindex_build_impl(c, ids ? ids : &none, n);
}
/* second index for the QUERY side of the pair stages: the listed reads' tables are read from it, the candidates' z-mers from the index of
 * wtz_zindex_build / _subset (which then only has to hold the reads this device sees as candidates).  ids == NULL && n == 0 drops it. */
extern "C" int wtz_zindex_build_queries(wtz_ctx_t *c, const uint32_t *ids, uint32_t n){
	if(!c) return wtz_fail(WTZ_E_ARG, "null argument");
	if(!ids && n) return wtz_fail(WTZ_E_ARG, "null argument");
	if(!ids){ c->zs[1].have = false; return WTZ_OK; }
	if(!c->zs[0].have) return wtz_fail(WTZ_E_STATE, "wtz_zindex_build_queries before wtz_zindex_build / wtz_zindex_build_subset");
	return zindex_build_impl(c, ids, n, 1);
}

/* ------------------------------------------------------------------------------------------------ */
/* A3: candidates                                                                                    */
/* ------------------------------------------------------------------------------------------------ */
/* asynchronous form: _begin uploads and launches on the context's stream and returns; _end waits and fetches.  Nothing else
 * may run on the context in between (the scratch pool is the kernel's); the host is free meanwhile. */
extern "C" int wtz_candidates_begin(wtz_ctx_t *c, const uint32_t *qids, uint32_t nq, const uint64_t *cand, const uint32_t *ncand_in){
	if(!c || !c->ktab || !qids || !cand || !ncand_in) return wtz_fail(WTZ_E_ARG, "index not built / null argument");
	if(c->cq_pending) return wtz_fail(WTZ_E_STATE, "wtz_candidates_begin: a request is already in flight");
	c->cq_n = nq; c->cq_groups = false;
	if(nq == 0){ c->cq_pending = true; return WTZ_OK; }
	CTX_ENTER(c);
	for(uint32_t i = 0; i < nq; i++) if(qids[i] >= c->n_reads) return wtz_fail(WTZ_E_ARG, "query id %u out of range", qids[i]);
	CHK(pool_reset(c));
	const uint32_t stride = c->P.ncand + 1;
	if(nq > c->cq_cap){
		(void)dev_sync();
		dev_free_persist(c->cq_q); dev_free_persist(c->cq_nc); dev_free_persist(c->cq_cand); dev_free_persist(c->cq_bytes); dev_free_persist(c->cq_thr);
		uint32_t cap = c->cq_cap ? c->cq_cap : 1024; while(cap < nq) cap *= 2;
		CHK(dev_alloc_persist((void**)&c->cq_q, (size_t)cap * 4)); CHK(dev_alloc_persist((void**)&c->cq_nc, (size_t)cap * 4)); CHK(dev_alloc_persist((void**)&c->cq_thr, (size_t)cap * 4));
		CHK(dev_alloc_persist((void**)&c->cq_cand, (size_t)cap * stride * 8)); CHK(dev_alloc_persist((void**)&c->cq_bytes, 8));
		c->cq_cap = cap;
	}
	uint32_t *d_q = c->cq_q, *d_n = c->cq_nc; uint64_t *d_cand = c->cq_cand; unsigned long long *d_bytes = c->cq_bytes;
	CHK(dev_h2d(d_q, qids, (size_t)nq * 4)); CHK(dev_h2d(d_n, ncand_in, (size_t)nq * 4)); CHK(dev_h2d(d_cand, cand, (size_t)nq * stride * 8));
	CHK(dev_set(d_bytes, 0, 8));
	const uint32_t *d_thr = NULL; (void)d_thr;
	if(c->idx_len_sorted && c->idx_end > c->idx_beg){
		/* per query: the first indexed read that is NOT longer than 1.2 x the query (lengths are non-increasing in the id) */
		std::vector<uint32_t> thr(nq);
		for(uint32_t i = 0; i < nq; i++){
			const uint32_t up = (uint32_t)(c->h_rdlen[qids[i]] * 1.2);              /* double multiply, wtzmo.c:445 */
			uint32_t lo = c->idx_beg, hi = c->idx_end;
			while(lo < hi){ const uint32_t mid = lo + (hi - lo) / 2; if(c->h_rdlen[mid] > up) lo = mid + 1; else hi = mid; }
			thr[i] = lo;
		}
		CHK(dev_h2d(c->cq_thr, thr.data(), (size_t)nq * 4)); d_thr = c->cq_thr;
	}
	const wtz_reads_t R = ctx_reads(c); const wtz_params_t *dP = c->dP; const wtz_kslot_t *tab = c->ktab; const uint64_t kmask = c->kmask;
	const uint32_t *seeds = c->kseeds; wtz_pool_t *pool = c->dpool;
	STAGE(c, "K_candidates");
	c->cq_tm.start();
	if(c->env_cand_wg && !c->env_cand_stream){
		/* one workgroup per query: partition by target read, sort each bucket in LDS (wtz_task_candidates_wg) */
		const uint32_t key_hi = c->idx_end << 1;
#ifdef WTZ_EMUL
		static thread_local uint32_t emul_cwg_lds[WTZ_CWG_LDS_BYTES / 4 + 16];
		uint32_t *lds_emul = emul_cwg_lds;
		CHK(wtz_launch_wg<K_candidates_wg>(nq, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_candidates_wg((uint32_t)t, R, d_q, dP, tab, kmask, seeds, pool, d_cand, d_n, stride, d_bytes, lds_emul, d_thr, key_hi); }, 1u, 0u));
#else
		CHK(wtz_launch_wg<K_candidates_wg>(nq, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_candidates_wg((uint32_t)t, R, d_q, dP, tab, kmask, seeds, pool, d_cand, d_n, stride, d_bytes, (uint32_t*)wtz_wave_scratch(), d_thr, key_hi); }, WTZ_CWG_THREADS, WTZ_CWG_LDS_BYTES));
#endif
	} else
#ifndef WTZ_EMUL
	{
		/* LDS per wave: the group table + output list + heap row of the streaming form (the sorting form of a query with too many groups
		 * uses the largest power-of-two window inside it) */
		uint32_t lds_b = c->env_cand_stream ? (WTZ_CAND_STREAM_LDS_BYTES(c->P.ncand) + 15u) & ~15u : WTZ_CAND_LDS_BYTES;
		if(lds_b < WTZ_CAND_LDS_BYTES || lds_b > 64u * 1024u) lds_b = WTZ_CAND_LDS_BYTES;
		if(lds_b != WTZ_CAND_LDS_BYTES) CHK(wtz_launch_coop<K_candidates_stream>(0, nq, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_candidates<true>((uint32_t)t, R, d_q, dP, tab, kmask, seeds, pool, d_cand, d_n, stride, d_bytes, (uint64_t*)wtz_wave_scratch(), lds_b / 8, d_thr); }, lds_b));
		/* the LDS window of the sorting form is a COMPILE-TIME constant: as a run-time value the windowed bitonic network loses its constant strides (26 -> 34 ms) */
		else CHK(wtz_launch_coop<K_candidates>(0, nq, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_candidates<false>((uint32_t)t, R, d_q, dP, tab, kmask, seeds, pool, d_cand, d_n, stride, d_bytes, (uint64_t*)wtz_wave_scratch(), WTZ_CAND_LDS_BYTES / 8, (const uint32_t*)NULL); }, WTZ_CAND_LDS_BYTES));
	}
#else
	CHK(wtz_launch_coop<K_candidates>(0, nq, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_candidates<false>((uint32_t)t, R, d_q, dP, tab, kmask, seeds, pool, d_cand, d_n, stride, d_bytes, (uint64_t*)NULL, 0); }));
#endif
	c->cq_tm.lap();
	c->cq_pending = true;
	return WTZ_OK;
}
extern "C" int wtz_candidates_end(wtz_ctx_t *c, uint64_t *cand, uint32_t *ncand_out){
	if(!c || !c->cq_pending) return wtz_fail(WTZ_E_STATE, "wtz_candidates_end without wtz_candidates_begin");
	c->cq_pending = false;
	const uint32_t nq = c->cq_n;
	if(nq == 0) return WTZ_OK;
	if(!cand || !ncand_out) return wtz_fail(WTZ_E_ARG, "null argument");
	CTX_ENTER(c);
	const uint32_t stride = c->P.ncand + 1;
	CHK(dev_sync());
	c->cnt.ms_candidates += c->cq_tm.read(); c->cnt.n_candidates_q += nq;
	{ unsigned long long hb = 0; CHK(dev_d2h(&hb, c->cq_bytes, 8)); c->cnt.bytes_seed_algo += hb; }
	CHK(dev_d2h(cand, c->cq_cand, (size_t)nq * stride * 8)); CHK(dev_d2h(ncand_out, c->cq_nc, (size_t)nq * 4));
	CHK(pool_check(c, "wtz_candidates"));
	return WTZ_OK;
}

/* A3 against a SHARD of the index: the (read, strand) groups with ol >= -d of every query, for the caller to join over the shards */
extern "C" int wtz_candidate_groups_begin(wtz_ctx_t *c, const uint32_t *qids, uint32_t nq){
	if(!c || !c->ktab || (nq && !qids)) return wtz_fail(WTZ_E_ARG, "index not built / null argument");
	if(c->cq_pending) return wtz_fail(WTZ_E_STATE, "wtz_candidate_groups_begin: a request is already in flight");
	c->cq_n = nq; c->cq_groups = true;
	if(nq == 0){ c->cq_pending = true; return WTZ_OK; }
	CTX_ENTER(c);
	for(uint32_t i = 0; i < nq; i++) if(qids[i] >= c->n_reads) return wtz_fail(WTZ_E_ARG, "query id %u out of range", qids[i]);
	CHK(pool_reset(c));
	const uint32_t stride = c->P.ncand + 1;
	if(nq > c->cq_cap){
		(void)dev_sync();
		dev_free_persist(c->cq_q); dev_free_persist(c->cq_nc); dev_free_persist(c->cq_cand); dev_free_persist(c->cq_bytes); dev_free_persist(c->cq_thr);
		uint32_t cap = c->cq_cap ? c->cq_cap : 1024; while(cap < nq) cap *= 2;
		CHK(dev_alloc_persist((void**)&c->cq_q, (size_t)cap * 4)); CHK(dev_alloc_persist((void**)&c->cq_nc, (size_t)cap * 4)); CHK(dev_alloc_persist((void**)&c->cq_thr, (size_t)cap * 4));
		CHK(dev_alloc_persist((void**)&c->cq_cand, (size_t)cap * stride * 8)); CHK(dev_alloc_persist((void**)&c->cq_bytes, 8));
		c->cq_cap = cap;
	}
	if(nq > c->cq_gcap){ (void)dev_sync(); dev_free_persist(c->cq_gptr); c->cq_gptr = NULL; uint32_t cap = c->cq_gcap ? c->cq_gcap : 1024; while(cap < nq) cap *= 2; CHK(dev_alloc_persist((void**)&c->cq_gptr, (size_t)cap * 8)); c->cq_gcap = cap; }
	uint32_t *d_q = c->cq_q, *d_n = c->cq_nc; uint64_t *d_cand = c->cq_cand, *d_gptr = c->cq_gptr; unsigned long long *d_bytes = c->cq_bytes;
	CHK(dev_h2d(d_q, qids, (size_t)nq * 4)); CHK(dev_set(d_n, 0, (size_t)nq * 4)); CHK(dev_set(d_gptr, 0, (size_t)nq * 8)); CHK(dev_set(d_bytes, 0, 8));
	const uint32_t *d_thr = NULL;
	if(c->idx_len_sorted && c->idx_end > c->idx_beg){
		std::vector<uint32_t> thr(nq);
		for(uint32_t i = 0; i < nq; i++){
			const uint32_t up = (uint32_t)(c->h_rdlen[qids[i]] * 1.2);              /* double multiply, wtzmo.c:445 */
			uint32_t lo = c->idx_beg, hi = c->idx_end;
			while(lo < hi){ const uint32_t mid = lo + (hi - lo) / 2; if(c->h_rdlen[mid] > up) lo = mid + 1; else hi = mid; }
			thr[i] = lo;
		}
		CHK(dev_h2d(c->cq_thr, thr.data(), (size_t)nq * 4)); d_thr = c->cq_thr;
	}
	const wtz_reads_t R = ctx_reads(c); const wtz_params_t *dP = c->dP; const wtz_kslot_t *tab = c->ktab; const uint64_t kmask = c->kmask;
	const uint32_t *seeds = c->kseeds; wtz_pool_t *pool = c->dpool; const uint32_t key_hi = c->idx_end << 1;
	STAGE(c, "K_candidates (groups)");
	c->cq_tm.start();
#ifdef WTZ_EMUL
	static thread_local uint32_t emul_cwg_lds[WTZ_CWG_LDS_BYTES / 4 + 16];
	uint32_t *lds_emul = emul_cwg_lds;
	CHK(wtz_launch_wg<K_candidates_wg>(nq, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_candidates_wg((uint32_t)t, R, d_q, dP, tab, kmask, seeds, pool, d_cand, d_n, stride, d_bytes, lds_emul, d_thr, key_hi, d_gptr); }, 1u, 0u));
#else
	CHK(wtz_launch_wg<K_candidates_wg>(nq, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_candidates_wg((uint32_t)t, R, d_q, dP, tab, kmask, seeds, pool, d_cand, d_n, stride, d_bytes, (uint32_t*)wtz_wave_scratch(), d_thr, key_hi, d_gptr); }, WTZ_CWG_THREADS, WTZ_CWG_LDS_BYTES));
#endif
	c->cq_tm.lap();
	c->cq_pending = true;
	return WTZ_OK;
}
extern "C" int wtz_candidate_groups_end(wtz_ctx_t *c, uint32_t *ngroups){
	if(!c || !c->cq_pending || !c->cq_groups) return wtz_fail(WTZ_E_STATE, "wtz_candidate_groups_end without wtz_candidate_groups_begin");
	c->cq_pending = false;
	const uint32_t nq = c->cq_n;
	c->cq_ng.assign(nq, 0);
	if(nq == 0) return WTZ_OK;
	if(!ngroups) return wtz_fail(WTZ_E_ARG, "null argument");
	CTX_ENTER(c);
	CHK(dev_sync());
	c->cnt.ms_candidates += c->cq_tm.read(); c->cnt.n_candidates_q += nq;
	{ unsigned long long hb = 0; CHK(dev_d2h(&hb, c->cq_bytes, 8)); c->cnt.bytes_seed_algo += hb; }
	CHK(dev_d2h(c->cq_ng.data(), c->cq_nc, (size_t)nq * 4));
	CHK(pool_check(c, "wtz_candidate_groups"));
	for(uint32_t i = 0; i < nq; i++){ if(c->cq_ng[i] == 0xFFFFFFFFu) return wtz_fail(WTZ_E_POOL, "wtz_candidate_groups: query %u ran out of scratch", i); ngroups[i] = c->cq_ng[i]; }
	return WTZ_OK;
}
extern "C" int wtz_candidate_groups_fetch(wtz_ctx_t *c, uint64_t *groups, uint64_t total){
	if(!c || !c->cq_groups) return wtz_fail(WTZ_E_STATE, "wtz_candidate_groups_fetch before wtz_candidate_groups_end");
	c->cq_groups = false;
	const uint32_t nq = c->cq_n;
	uint64_t tot = 0; for(uint32_t i = 0; i < nq; i++) tot += c->cq_ng[i];
	if(tot != total) return wtz_fail(WTZ_E_ARG, "wtz_candidate_groups_fetch: expected room for %llu groups, got %llu", (unsigned long long)tot, (unsigned long long)total);
	if(tot == 0) return WTZ_OK;
	if(!groups) return wtz_fail(WTZ_E_ARG, "null output");
	CTX_ENTER(c);
	std::vector<uint64_t> off((size_t)nq + 1);
	uint64_t o = 0; for(uint32_t i = 0; i < nq; i++){ off[i] = o; o += c->cq_ng[i]; } off[nq] = o;
	uint64_t *d_off = NULL, *d_g = NULL;
	CHK(dev_alloc((void**)&d_off, off.size() * 8)); CHK(dev_h2d(d_off, off.data(), off.size() * 8));
	CHK(dev_alloc((void**)&d_g, (size_t)tot * 8));
	const uint64_t *gp = c->cq_gptr;
	CHK(wtz_launch<K_pack_groups>(0, nq, [=] WTZ_LAMBDA (uint64_t t){ const uint64_t *src = (const uint64_t*)(uintptr_t)gp[t]; const uint64_t n = d_off[t + 1] - d_off[t]; for(uint64_t k = 0; k < n; k++) d_g[d_off[t] + k] = src[k]; }));
	CHK(dev_sync());
	CHK(dev_d2h(groups, d_g, (size_t)tot * 8));
	dev_free(d_off); dev_free(d_g);
	return WTZ_OK;
}
extern "C" int wtz_candidates(wtz_ctx_t *c, const uint32_t *qids, uint32_t nq, uint64_t *cand, uint32_t *ncand_io){
	if(!c || !c->ktab || !qids || !cand || !ncand_io) return wtz_fail(WTZ_E_ARG, "index not built / null argument");
	if(nq == 0) return WTZ_OK;
	int rc = wtz_candidates_begin(c, qids, nq, cand, ncand_io);
	if(rc != WTZ_OK) return rc;
	return wtz_candidates_end(c, cand, ncand_io);
}

/* ------------------------------------------------------------------------------------------------ */
/* per-batch pair stages                                                                             */
/* ------------------------------------------------------------------------------------------------ */
extern "C" int wtz_batch_begin(wtz_ctx_t *c){
	if(!c) return wtz_fail(WTZ_E_ARG, "null context");
	CTX_ENTER(c);
	free_batch(c);
	return pool_reset(c);
}

#if defined(WTZ_DEBUG_CRUMBS) && !defined(WTZ_EMUL)
#include <signal.h>
#include <unistd.h>
static unsigned int *g_crumbs = NULL; static uint32_t g_crumbs_n = 0; static const uint32_t *g_crumbs_q = NULL, *g_crumbs_c = NULL;
static void wtz_crumbs_dump(int sig){
	unsigned hist[256]; memset(hist, 0, sizeof hist); unsigned shown = 0;
	for(uint32_t i = 0; i < g_crumbs_n; i++) hist[g_crumbs[i] & 0xFF]++;
	fprintf(stderr, "[crumbs] signal %d, %u pairs; tasks per last point:", sig, g_crumbs_n);
	for(int k = 0; k < 256; k++) if(hist[k]) fprintf(stderr, " %d:%u", k, hist[k]);
	fprintf(stderr, "\n");
	for(uint32_t i = 0; i < g_crumbs_n && shown < 16; i++) if((g_crumbs[i] & 0xFF) != 0xFF && (g_crumbs[i] & 0xFF) != 0){ fprintf(stderr, "[crumbs]   pair %u (q %u, c %u): point %u, hits %u\n", i, g_crumbs_q[i], g_crumbs_c[i], g_crumbs[i] & 0xFF, g_crumbs[i] >> 8); shown++; }
	fflush(stderr); _exit(86);
}
#endif

extern "C" int wtz_pairs_seed(wtz_ctx_t *c, const uint32_t *qid, const uint32_t *cid, uint32_t n, wtz_pair_summary_t *out){
	if(!c || !c->zs[0].have || (n && (!qid || !cid || !out))) return wtz_fail(WTZ_E_ARG, "z-index not built / null argument");
	CTX_ENTER(c);
	free_batch(c);
	CHK(pool_reset(c));
	if(n == 0){ c->n_pairs = 0; c->h_pairres.clear(); c->have_pairs = true; return WTZ_OK; }
	for(uint32_t i = 0; i < n; i++) if(qid[i] >= c->n_reads || cid[i] >= c->n_reads) return wtz_fail(WTZ_E_ARG, "pair %u: read id out of range", i);
	CHK(reserve_pairs(c, n));
	CHK(dev_h2d(c->d_qid, qid, (size_t)n * 4)); CHK(dev_h2d(c->d_cid, cid, (size_t)n * 4));
	const wtz_env_t V = ctx_env(c); const uint32_t *dq = c->d_qid, *dc = c->d_cid; wtz_pairres_t *dr = c->d_pairres;
	wtz_timer tm; tm.start();
	wtz_timer t1; t1.start();
	STAGE(c, "K_pair");
#if defined(WTZ_DEBUG_CRUMBS) && !defined(WTZ_EMUL)
	unsigned int *h_crumbs = NULL;
	if(getenv("WTZ_DEBUG_CRUMBS")){
		HIPCHK(hipHostMalloc((void**)&h_crumbs, (size_t)n * 4, hipHostMallocCoherent | hipHostMallocMapped)); memset(h_crumbs, 0, (size_t)n * 4);
		unsigned int *dptr = NULL; HIPCHK(hipHostGetDevicePointer((void**)&dptr, h_crumbs, 0));
		HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(wtz_crumbs), &dptr, sizeof dptr));
		g_crumbs = h_crumbs; g_crumbs_n = n; g_crumbs_q = qid; g_crumbs_c = cid;
		signal(SIGABRT, wtz_crumbs_dump); signal(SIGPIPE, wtz_crumbs_dump); signal(SIGSEGV, wtz_crumbs_dump); signal(SIGBUS, wtz_crumbs_dump); signal(SIGTERM, wtz_crumbs_dump);
	}
#endif
	/* XCD-aware task order: workgroups are dealt round-robin over the 8 XCDs, each with its own L2, and the pairs of a range are listed query by
	 * query (about 30 candidates each), all of them searching the same query-side z-mer tables.  With the identity mapping an XCD's ~640 resident
	 * waves hold every 8th pair of a 5 000-pair stretch, i.e. the tables of ~170 queries (20 MB against 4 MB of L2); giving every XCD runs of
	 * `xg` CONSECUTIVE pairs makes that ~25 queries.  (WTZ_XCD_GROUP=0: identity.) */
	const uint32_t xg = c->env_xcd_group;
	/* Heavy pairs first (round 4).  A pair's work grows faster than linearly with its matches (~ len(q) * len(c) / 78 732 chance matches of 10-mers alone), and
	 * once the average dmo pair took a few ms the launch of a range ended with ONE wave still on a pair of 6 000 - 24 000 matches (100 - 230 M cycles of a
	 * 120 - 170 ms launch: phase profile).  The n / 32 pairs with the largest len(q) * len(c) therefore head the task order (largest first); the others keep
	 * the plan order - consecutive pairs share their query's tables - under the XCD mapping below.  WTZ_PAIR_HEAVY_FIRST=0 / 1 overrides the engine default
	 * (dmo on; zmo off: its launches were measured full to the end). */
	uint32_t nh = 0; const uint32_t *d_ord = NULL;
#ifndef WTZ_EMUL
	if(c->env_heavy_first >= 0 ? c->env_heavy_first != 0 : c->P.dot_matrix != 0){
		nh = n / 32u;
		if(nh >= 8u){
			std::vector<uint64_t> key(n); std::vector<uint32_t> ord(n), hv(n);
			for(uint32_t i = 0; i < n; i++){ key[i] = (uint64_t)c->h_rdlen[qid[i]] * c->h_rdlen[cid[i]]; hv[i] = i; }
			std::nth_element(hv.begin(), hv.begin() + nh, hv.end(), [&](uint32_t a, uint32_t b){ return key[a] != key[b] ? key[a] > key[b] : a < b; });
			std::sort(hv.begin(), hv.begin() + nh, [&](uint32_t a, uint32_t b){ return key[a] != key[b] ? key[a] > key[b] : a < b; });
			std::vector<uint8_t> heavy(n, 0);
			for(uint32_t k = 0; k < nh; k++){ ord[k] = hv[k]; heavy[hv[k]] = 1; }
			uint32_t w = nh; for(uint32_t i = 0; i < n; i++) if(!heavy[i]) ord[w++] = i;
			uint32_t *dd = NULL; CHK(dev_alloc((void**)&dd, (size_t)n * 4)); CHK(dev_h2d(dd, ord.data(), (size_t)n * 4)); d_ord = dd;
		} else nh = 0;
	}
#endif
	const uint64_t n64 = n - nh; const uint64_t nh64 = nh;
#ifdef WTZ_EMUL
	CHK(wtz_launch_coop<K_pair>(0, n, [=] WTZ_LAMBDA (uint64_t b){ (void)xg; (void)n64; (void)nh64; (void)d_ord; wtz_task_pair<-1>((uint32_t)b, V, dq, dc, dr); }, c->P.dot_matrix ? WTZ_PAIR_DM_LDS_BYTES : WTZ_PAIR_LDS_BYTES));
#else
	if(c->P.dot_matrix){
		CHK(wtz_launch_coop<K_pair_dm>(0, n, [=] WTZ_LAMBDA (uint64_t b){
			uint64_t t = b;
			if(b >= nh64){ const uint64_t b2 = b - nh64; t = b2; if(xg){ const uint64_t per = 8ull * xg, full = n64 / per * per; if(b2 < full){ const uint64_t r = b2 % per; t = b2 - r + (r & 7u) * xg + (r >> 3); } } t += nh64; }
			if(d_ord) t = d_ord[t];
			wtz_task_pair<1>((uint32_t)t, V, dq, dc, dr); }, WTZ_PAIR_DM_LDS_BYTES));
	} else {
		CHK(wtz_launch_coop<K_pair>(0, n, [=] WTZ_LAMBDA (uint64_t b){
			uint64_t t = b;
			if(b >= nh64){ const uint64_t b2 = b - nh64; t = b2; if(xg){ const uint64_t per = 8ull * xg, full = n64 / per * per; if(b2 < full){ const uint64_t r = b2 % per; t = b2 - r + (r & 7u) * xg + (r >> 3); } } t += nh64; }
			if(d_ord) t = d_ord[t];
#ifdef WTZ_PAIR_TWO_LAUNCH
			wtz_task_pair<0, false>((uint32_t)t, V, dq, dc, dr); }, WTZ_PAIR_LDS_BYTES));       /* experiment: pairs with ranges beyond the LDS slice are marked and finished by K_pair_zbig */
#else
			wtz_task_pair<0, true>((uint32_t)t, V, dq, dc, dr); }, WTZ_PAIR_LDS_BYTES));
#endif
	}
#endif
#if defined(WTZ_DEBUG_CRUMBS) && !defined(WTZ_EMUL)
	if(h_crumbs){
		const double t0 = wtz_wall(); const double limit = atof(getenv("WTZ_DEBUG_CRUMBS")) > 1 ? atof(getenv("WTZ_DEBUG_CRUMBS")) : 20.0;
		while(hipStreamQuery(g_stream) == hipErrorNotReady && wtz_wall() - t0 < limit){ struct timespec ts = {0, 50000000}; nanosleep(&ts, NULL); }
		if(hipStreamQuery(g_stream) == hipErrorNotReady){
			unsigned hist[256]; memset(hist, 0, sizeof hist); unsigned shown = 0;
			for(uint32_t i = 0; i < n; i++) hist[h_crumbs[i] & 0xFF]++;
			fprintf(stderr, "[crumbs] K_pair still running after %.0f s, %u pairs; tasks per last point:", limit, n);
			for(int k = 0; k < 256; k++) if(hist[k]) fprintf(stderr, " %d:%u", k, hist[k]);
			fprintf(stderr, "\n");
			for(uint32_t i = 0; i < n && shown < 12; i++) if((h_crumbs[i] & 0xFF) != 0xFF && (h_crumbs[i] & 0xFF) != 0){ fprintf(stderr, "[crumbs]   pair %u (q %u, c %u): point %u, hits %u\n", i, qid[i], cid[i], h_crumbs[i] & 0xFF, h_crumbs[i] >> 8); shown++; }
			fflush(stderr); _exit(86);
		}
	}
#endif
	CHK(dev_sync());
	{ const double ms1 = t1.stop(); if(c->env_profile) fprintf(stderr, "[pair-profile] K_pair first launch: %u pairs, %.1f ms\n", n, ms1); }
	c->n_pairs = n; c->h_pairres.resize(n); c->have_pairs = true;
	CHK(dev_d2h(c->h_pairres.data(), c->d_pairres, (size_t)n * sizeof(wtz_pairres_t)));
#ifndef WTZ_EMUL
	if(!c->P.dot_matrix){
		/* zmo pairs with a window range that does not fit the LDS slice (hundreds of matches of one strand inside one window: repeats) were left by the first launch:
		 * the launch that carries the pool-workspace body of the scan finishes them (round 3 ran those scans on lane 0 and the heaviest pair bounded its launch) */
		std::vector<uint32_t> list;
		for(uint32_t i = 0; i < n; i++) if(c->h_pairres[i].gate && c->h_pairres[i].dm_dir == WTZ_PAIR_NEEDS_ZBIG && !c->h_pairres[i].bad) list.push_back(i);
		if(!list.empty()){
			uint32_t *d_list = NULL; CHK(dev_alloc((void**)&d_list, list.size() * 4)); CHK(dev_h2d(d_list, list.data(), list.size() * 4));
			wtz_timer tt; tt.start();
			STAGE(c, "K_pair_zbig");
			CHK(wtz_launch_coop<K_pair_zbig>(0, list.size(), [=] WTZ_LAMBDA (uint64_t t){ wtz_task_pair<0, true>(d_list[t], V, dq, dc, dr); }, WTZ_PAIR_LDS_BYTES));
			CHK(dev_sync());
			const double ms_t = tt.stop();
			dev_free(d_list);
			CHK(dev_d2h(c->h_pairres.data(), c->d_pairres, (size_t)n * sizeof(wtz_pairres_t)));
			if(c->env_profile) fprintf(stderr, "[pair-profile] zmo pairs with ranges beyond the LDS slice: %zu of %u, %.1f ms\n", list.size(), n, ms_t);
		}
	}
#endif
	if(c->P.dot_matrix){
		/* pairs whose strand images exceed the LDS slice of K_pair are finished by launches with larger slices: few pairs,
		 * but they are the long ones that would otherwise bound the batch from a single lane */
		uint32_t tiers[3] = { WTZ_PAIR_DM_LDS_TIER2, WTZ_PAIR_DM_LDS_TIER3, WTZ_PAIR_DM_LDS_TIER4 };
		if(getenv("WTZ_DM_TIER3_KB")) tiers[1] = (uint32_t)atoi(getenv("WTZ_DM_TIER3_KB")) << 10;
		if(getenv("WTZ_DM_TIER4_KB")) tiers[2] = (uint32_t)atoi(getenv("WTZ_DM_TIER4_KB")) << 10;
		/* with the pool image allowed in the first launch only what overflowed its group table or band list is left: the last launch's */
		for(int tier = c->env_dm_first_big ? 2 : 0; tier < 3; tier++){
			std::vector<uint32_t> list;
			for(uint32_t i = 0; i < n; i++) if(c->h_pairres[i].gate && c->h_pairres[i].dm_dir == -2 && !c->h_pairres[i].bad) list.push_back(i);
			if(list.empty()) break;
			uint32_t *d_list = NULL; CHK(dev_alloc((void**)&d_list, list.size() * 4)); CHK(dev_h2d(d_list, list.data(), list.size() * 4));
			const uint32_t lb = tiers[tier]; const bool last = (tier == 2), big = (tier >= 1);
			wtz_timer tt; tt.start();
			STAGE(c, "K_pair_big");
			CHK(wtz_launch_coop<K_pair_big>(0, list.size(), [=] WTZ_LAMBDA (uint64_t t){ wtz_task_pair_dm_big((uint32_t)t, V, d_list, dq, dc, dr, lb, last, big); }, lb));
			CHK(dev_sync());
			const double ms_t = tt.stop();
			dev_free(d_list);
			CHK(dev_d2h(c->h_pairres.data(), c->d_pairres, (size_t)n * sizeof(wtz_pairres_t)));
			if(c->env_profile) fprintf(stderr, "[pair-profile] dmo tier %d (%u KB LDS): %zu pairs, %.1f ms\n", tier + 2, lb >> 10, list.size(), ms_t);
		}
	}
	c->cnt.ms_pairs += tm.stop(); c->cnt.n_pairs += n;
	for(uint32_t i = 0; i < n; i++) c->cnt.bytes_zmer_algo += (uint64_t)c->h_rdlen[cid[i]] / 4 + 16ull * c->h_pairres[i].n_hits;
	CHK(pool_check(c, "wtz_pairs_seed"));
	if(c->env_profile){
		uint64_t sum[4] = {0, 0, 0, 0}; uint32_t mx[4] = {0, 0, 0, 0}, arg = 0;
		for(uint32_t i = 0; i < n; i++){ for(int k = 0; k < 4; k++){ sum[k] += c->h_pairres[i].tick[k]; if(c->h_pairres[i].tick[k] > mx[k]){ mx[k] = c->h_pairres[i].tick[k]; if(k == 3) arg = i; } } }
		fprintf(stderr, "[pair-profile] n=%u kticks sum match/sort/win/total %llu/%llu/%llu/%llu max %u/%u/%u/%u; slowest pair: hits %u (its match/sort/win %u/%u/%u)\n", n,
			(unsigned long long)sum[0], (unsigned long long)sum[1], (unsigned long long)sum[2], (unsigned long long)sum[3], mx[0], mx[1], mx[2], mx[3],
			c->h_pairres[arg].n_hits, c->h_pairres[arg].tick[0], c->h_pairres[arg].tick[1], c->h_pairres[arg].tick[2]);
	}
	for(uint32_t i = 0; i < n; i++){
		const wtz_pairres_t &r = c->h_pairres[i];
		if(r.bad) return wtz_fail(WTZ_E_POOL, "wtz_pairs_seed: pair %u ran out of scratch", i);
		wtz_pair_summary_t s; memset(&s, 0, sizeof s);
		s.n_hits = r.n_hits; s.gate = r.gate; s.ovl[0] = r.ovl[0]; s.ovl[1] = r.ovl[1]; s.nwin[0] = r.nwin[0]; s.nwin[1] = r.nwin[1];
		s.dm_score = r.dm_score; s.dm_qb = r.dm_qb; s.dm_qe = r.dm_qe; s.dm_tb = r.dm_tb; s.dm_te = r.dm_te; s.dm_dir = r.dm_dir;
		out[i] = s;
	}
	return WTZ_OK;
}

extern "C" int wtz_pairs_windows(wtz_ctx_t *c, wtz_winbox_t *wins, uint64_t n_wins){
	if(!c || !c->have_pairs) return wtz_fail(WTZ_E_STATE, "wtz_pairs_windows before wtz_pairs_seed");
	CTX_ENTER(c);
	uint64_t tot = 0; for(uint32_t i = 0; i < c->n_pairs; i++) tot += c->h_pairres[i].nwin[0] + c->h_pairres[i].nwin[1];
	if(tot != n_wins) return wtz_fail(WTZ_E_ARG, "wtz_pairs_windows: expected room for %llu windows, got %llu", (unsigned long long)tot, (unsigned long long)n_wins);
	if(tot == 0) return WTZ_OK;
	if(!wins) return wtz_fail(WTZ_E_ARG, "null output");
	std::vector<uint64_t> off((size_t)c->n_pairs * 2 + 1);
	uint64_t o = 0; for(uint32_t i = 0; i < c->n_pairs; i++) for(int d = 0; d < 2; d++){ off[(size_t)i * 2 + d] = o; o += c->h_pairres[i].nwin[d]; }
	off[(size_t)c->n_pairs * 2] = o;
	uint64_t *d_off = NULL; wtz_winbox_t *d_w = NULL;
	CHK(dev_alloc((void**)&d_off, off.size() * 8)); CHK(dev_h2d(d_off, off.data(), off.size() * 8));
	CHK(dev_alloc((void**)&d_w, (size_t)tot * sizeof(wtz_winbox_t)));
	const wtz_pairres_t *dr = c->d_pairres;
	CHK(wtz_launch<K_pack_windows>(0, (uint64_t)c->n_pairs * 2, [=] WTZ_LAMBDA (uint64_t t){
		const wtz_pairres_t &r = dr[t >> 1]; const uint32_t d = (uint32_t)(t & 1);
		for(uint32_t k = 0; k < r.nwin[d]; k++){ wtz_winbox_t b; b.beg[0] = r.win[d][k].beg[0]; b.beg[1] = r.win[d][k].beg[1]; b.end[0] = r.win[d][k].end[0]; b.end[1] = r.win[d][k].end[1]; d_w[d_off[t] + k] = b; }
	}));
	CHK(dev_sync());
	CHK(dev_d2h(wins, d_w, (size_t)tot * sizeof(wtz_winbox_t)));
	dev_free(d_off); dev_free(d_w);
	return WTZ_OK;
}

#ifndef WTZ_EMUL
/* upper bound of the transient-pool bytes one K-sw3 job takes (trace rows come 64 at a time; a row is the widest of the wave forms), its row bound and its band class */
WTZ_HD uint64_t wtz_ext_trace_need(int32_t qlen, int32_t tlen, int32_t init, int32_t W, int32_t M, int32_t O, int32_t E, int32_t T, int32_t *ql_out, int32_t *ncol_out){
	if(ql_out) *ql_out = 0;
	if(ncol_out) *ncol_out = 0;
	if(qlen <= 0 || tlen <= 0) return 0;
	if(init < 0) init = 0;
	int32_t ql, tl, n_col;
	wtz_ext_geometry(qlen, tlen, init, W, M, O, O, E, T, ql, tl, n_col);
	if(ql_out) *ql_out = ql;
	if(ncol_out) *ncol_out = n_col;
	/* one-wave register kernel (4-column steps), four-wave kernel (256 lanes), LDS-ring kernel (odd columns per lane) */
	const uint64_t c_reg = ((uint64_t)(n_col + 63) / 64 + 3) / 4, c_mw = ((uint64_t)(n_col + 255) / 256 + 3) / 4 * 4, c_gen = ((((uint64_t)(n_col + 63) / 64) | 1) + 3) / 4;
	uint64_t zrow = (c_reg > c_gen ? c_reg : c_gen) * 256; if(c_mw * 256 > zrow) zrow = c_mw * 256;
	uint64_t nb = ((uint64_t)(ql + 63) / 64) * 64 * zrow + (uint64_t)WTZ_TRACE_MAXCHUNK * 8 + (uint64_t)(ql + 2) * 4 + 256;
	if((n_col + 63) / 64 > 32 || (tl + 63) / 32 + 1 > 1032){      /* outside the wave forms: the scalar body's row arrays and byte matrix, grown in powers of two */
		uint64_t z = 1024; while(z < (uint64_t)ql * (uint64_t)n_col) z <<= 1;
		uint64_t r = 64; while(r < (uint64_t)tl + 3) r <<= 1;
		uint64_t zb = 64; while(zb < (uint64_t)ql + 2) zb <<= 1;
		nb = z + 8 * r + 4 * zb + 256;
	}
	return nb;
}
static uint64_t ext_trace_need(const wtz_ctx *c, int32_t qlen, int32_t tlen, int32_t init, int32_t W, int32_t *ql_out, int32_t *ncol_out){
	return wtz_ext_trace_need(qlen, tlen, init, W, c->P.M, c->P.O, c->P.E, c->P.T, ql_out, ncol_out);
}

/* Both end extensions of every item of the stage on one wavefront per item (wtz_stitch_fused.h).  The items are ordered by the rows their two extensions can
 * run at most (longest first) and the launch is made only if the traces of ALL jobs fit the transient pool together (their geometry is known before any
 * extension has run: wtz_task_stitch_left's rgeo); otherwise c->fused_ran stays false and the stage runs its launches one after the other as before. */
static int run_stitch_fused(wtz_ctx *c, const wtz_env_t &V, const wtz_alnitem_t *d_items, wtz_stitch_state_t *d_st, wtz_extjob_t *d_jl, wtz_extjob_t *d_jr, const wtz_gapres_t *d_gaps, const int32_t *d_rgeo, uint32_t m){
	c->fused_ran = false;
	if(m == 0) return WTZ_OK;
	/* order and budget on the device (the host form - fetch the geometry, order 31 000 items, send the order back - was 2.7 ms of an idle device per range):
	 * key = the rows both jobs can run at most, inverted (ascending stable radix sort = longest first, ties in item order); the trace bounds are summed with an atomic */
	uint64_t *d_k = NULL; uint32_t *d_order = NULL; unsigned long long *d_acc = NULL;
	uint32_t *d_open = NULL;
	CHK(dev_alloc((void**)&d_k, (size_t)m * 8)); CHK(dev_alloc((void**)&d_order, (size_t)m * 4)); CHK(dev_alloc((void**)&d_acc, 32)); CHK(dev_set(d_acc, 0, 32));
	if(c->env_ext_pk) CHK(dev_alloc((void**)&d_open, ((size_t)m + 1) * 4));
	{
		const int32_t pM = c->P.M, pO = c->P.O, pE = c->P.E, pT = c->P.T, pW = -c->P.ew;
		const uint32_t mw_rows = c->env_ext_mw_rows > 0 ? (uint32_t)c->env_ext_mw_rows : 0xFFFFFFFFu;
		const bool use_pk = c->env_ext_pk != 0; const wtz_params_t *dP = V.P;
		CHK(wtz_launch<K_misc>(0, m, [=] WTZ_LAMBDA (uint64_t t){
			const wtz_extjob_t &j = d_jl[t];
			int32_t qa = 0, qb = 0;
			unsigned long long nb = wtz_ext_trace_need(j.valid ? j.qlen : -1, j.tlen, 0, pW, pM, pO, pE, pT, &qa, (int32_t*)NULL);
			nb += wtz_ext_trace_need(d_rgeo[2 * t], d_rgeo[2 * t + 1], 0, pW, pM, pO, pE, pT, &qb, (int32_t*)NULL);
			const uint32_t rows = (uint32_t)qa + (uint32_t)qb;
			/* which form takes the item (bit 32 of the key: the items of the 32-bit form end up behind those of the packed form, both longest-first): the packed
			 * form needs both extensions inside its 16-bit window - the left one's init_score is known, the right one's is not (wtz_pk_window_any_init) */
			uint32_t to_fr = 0;
			if(use_pk){
				if(j.valid && j.qlen > 0 && j.tlen > 0){
					int32_t W = pW, ql = 0, tl = 0, nc = 0, bias, ng, sh; const int32_t in0 = j.init_score < 0 ? 0 : j.init_score;
					wtz_ext_geometry(j.qlen, j.tlen, in0, W, pM, pO, pO, pE, pT, ql, tl, nc);
					if(!wtz_pk_window(dP, in0, ql, tl, &bias, &ng, &sh)) to_fr = 1;
				}
				if(d_rgeo[2 * t] > 0 && d_rgeo[2 * t + 1] > 0){
					/* the right extension's init_score (wtz_task_stitch_mid) = the left extension's score - 100 M + the windows and gaps behind the first window: all of it
					 * known here but the left extension's gain, which lies in [0, M * min(its two sides)] */
					const wtz_stitch_state_t &st = d_st[t]; const wtz_alnitem_t &it = d_items[t];
					long long i_lo = st.x.score, gain = 0;
					if(j.valid && j.qlen > 0 && j.tlen > 0) gain = (long long)pM * (j.qlen < j.tlen ? j.qlen : j.tlen);
					const wtz_gapres_t *gp = d_gaps + (it.regs - d_items[0].regs);
					for(uint32_t k = st.first + 1; k < it.nwin; k++) if(it.regs[k].pass == 1) i_lo += (long long)gp[k].score + it.regs[k].x.score;
					int32_t W = pW, ql = 0, tl = 0, nc = 0;
					wtz_ext_geometry(d_rgeo[2 * t], d_rgeo[2 * t + 1], 0, W, pM, pO, pO, pE, pT, ql, tl, nc);
					if(!wtz_pk_window_range(dP, ql, tl, i_lo, i_lo + gain)) to_fr = 1;
				}
			}
			d_k[t] = ((uint64_t)to_fr << 32) | (uint64_t)(0xFFFFFFFFu - rows); d_order[t] = (uint32_t)t;
			if(to_fr) WTZ_ATOMIC_ADD64(&d_acc[3], 1ull);
			if(nb) WTZ_ATOMIC_ADD64(&d_acc[0], nb);
			if(rows) WTZ_ATOMIC_ADD64(&d_acc[1], (unsigned long long)rows);
			if(rows >= mw_rows) WTZ_ATOMIC_ADD64(&d_acc[2], 1ull);
		}));
	}
	CHK(dev_sort_pairs_u64_u32(d_k, d_order, m, 33));
	unsigned long long h_acc[4] = {0, 0, 0, 0}; CHK(dev_d2h(h_acc, d_acc, 32));
	const uint64_t acc = h_acc[0]; const unsigned long long ext_sum = h_acc[1];
	const uint64_t budget = (c->pool_bytes - c->main_bytes) / 16 * 15;
	/* acc sums UPPER bounds (every job run to its last row); the traces are allocated 64 rows at a time as a job runs, and most jobs end early: what the launches
	 * before this one took of their bounds (x 1.3, never below a fifth) is what this one is expected to take.  An estimate that was too low ends in WTZ_E_POOL like
	 * any other exhausted pool: the host redoes the range in halves. */
	/* what does not fit at once runs in up to four groups (every ng-th item of the order each: all groups are ordered longest-first), the transient pool reset between them */
	uint32_t ng = 1; while(ng < 4 && (double)acc * c->ext_use_ratio / ng > (double)budget) ng++;
	if((double)acc * c->ext_use_ratio / ng > (double)budget){ if(c->env_profile) fprintf(stderr, "[ext-profile] fused launch declined: %u items, trace bounds %.1f GB x %.2f against %.1f GB\n", m, acc / 1e9, c->ext_use_ratio, budget / 1e9); dev_free(d_order); dev_free(d_k); dev_free(d_acc); if(d_open) dev_free(d_open); return WTZ_OK; }          /* the two launches cut their jobs into groups that fit */
	double ms_l = 0; uint64_t used_sum = 0;
	for(uint32_t g = 0; g < ng; g++){
		const uint32_t mg = (m - g + ng - 1) / ng;
		if(mg == 0) continue;
		CHK(tpool_reset(c));
		wtz_timer te; te.start();
		/* the items at the head of the order (longest first) whose extensions can run >= WTZ_EXT_MW_ROWS rows: four wavefronts each on the side stream, beside the
		 * one-wavefront launch over the rest - they are the launch's critical path (a row takes a wavefront ~2 us whatever else the device does) */
		uint32_t n_long = 0;
		if(ng == 1 && c->env_ext_mw_rows > 0){ n_long = (uint32_t)h_acc[2]; if(n_long > m / 8u) n_long = m / 8u; }
		if(n_long){
			HIPCHK(hipEventRecord(c->ev_mw_fork, g_stream)); HIPCHK(hipStreamWaitEvent(c->stream_mw, c->ev_mw_fork, 0));
			hipLaunchKernelGGL((wtz_kernel_stitch_ext_frmw<1032>), dim3(n_long), dim3(256), WTZ_WAVE_LDS_BYTES, c->stream_mw, V, d_items, d_st, d_jl, d_jr, d_gaps, (const uint32_t*)d_order, n_long);
			HIPCHK(hipGetLastError());
			HIPCHK(hipEventRecord(c->ev_mw_join, c->stream_mw));
		}
		if(mg > n_long && c->env_ext_pk){
			/* the packed 16-bit form; the items dealt to the 32-bit form beforehand (the tail of the order: a handful of the longest extensions per step) run beside
			 * it on the side stream; what the packed form declines after all is listed and finished by the 32-bit form behind it */
			uint32_t n_fr = (ng == 1 && !n_long) ? (uint32_t)h_acc[3] : 0u;
			if(n_fr > mg - n_long) n_fr = mg - n_long;
			CHK(dev_set(d_open, 0, 4));
			if(n_fr){
				HIPCHK(hipEventRecord(c->ev_mw_fork, g_stream)); HIPCHK(hipStreamWaitEvent(c->stream_mw, c->ev_mw_fork, 0));
				hipLaunchKernelGGL((wtz_kernel_stitch_ext_fr<1032>), dim3(n_fr), dim3(64), WTZ_WAVE_LDS_BYTES, c->stream_mw, V, d_items, d_st, d_jl, d_jr, d_gaps, (const uint32_t*)d_order + (mg - n_fr), n_fr, 1u, 0u);
				HIPCHK(hipGetLastError());
				HIPCHK(hipEventRecord(c->ev_mw_join, c->stream_mw));
			}
			const uint32_t n_pk = mg - n_long - n_fr;
			if(n_pk){
				hipLaunchKernelGGL((wtz_kernel_stitch_ext_pk<1032>), dim3(n_pk), dim3(64), WTZ_PK_LDS_BYTES(1032), g_stream, V, d_items, d_st, d_jl, d_jr, d_gaps, (const uint32_t*)d_order + n_long, n_pk, ng, g, d_open);
				HIPCHK(hipGetLastError());
			}
			if(n_fr) HIPCHK(hipStreamWaitEvent(g_stream, c->ev_mw_join, 0));
			uint32_t n_open = 0; CHK(dev_d2h(&n_open, d_open, 4));
			c->ext_open_total += n_open; c->ext_fr_total += n_fr;
			if(n_open){ hipLaunchKernelGGL((wtz_kernel_stitch_ext_fr<1032>), dim3(n_open), dim3(64), WTZ_WAVE_LDS_BYTES, g_stream, V, d_items, d_st, d_jl, d_jr, d_gaps, (const uint32_t*)d_open + 1, n_open, 1u, 0u); HIPCHK(hipGetLastError()); }
		} else if(mg > n_long) hipLaunchKernelGGL((wtz_kernel_stitch_ext_fr<1032>), dim3(mg - n_long), dim3(64), WTZ_WAVE_LDS_BYTES, g_stream, V, d_items, d_st, d_jl, d_jr, d_gaps, (const uint32_t*)d_order + n_long, mg - n_long, ng, g);
		HIPCHK(hipGetLastError());
		if(n_long) HIPCHK(hipStreamWaitEvent(g_stream, c->ev_mw_join, 0));
		ms_l += te.stop();
		if(c->env_profile && n_long) fprintf(stderr, "[ext-profile] %u of %u items on four wavefronts each\n", n_long, m);
		{
			const int rc_t = tpool_check(c, "K-sw3 extension jobs (both ends on one wavefront)");
			if(rc_t != WTZ_OK){
				/* the budget under-estimated what the traces take: the next launch is planned with twice the share (the host redoes this range in halves and is told
				 * that it was the transient pool, so that its bytes-per-pair estimate of the MAIN pool is left alone: wtz_pool_failure_kind) */
				c->ext_use_ratio = c->ext_use_ratio * 2.0 > 1.0 ? 1.0 : c->ext_use_ratio * 2.0;
				dev_free(d_order); dev_free(d_k); dev_free(d_acc); if(d_open) dev_free(d_open);
				return rc_t;
			}
		}
		used_sum += c->tpool_last_used;
	}
	c->cnt.ms_ext += ms_l; c->cnt.n_extjobs += 2ull * m;
	c->fused_ran = true;
	if(c->env_profile) fprintf(stderr, "[ext-profile] fused launch: %u items in %u group(s), rows (upper bound) sum %llu, %.2f ms; items of the 32-bit form so far: dealt %llu, declined by the packed form %llu\n", m, ng, ext_sum, ms_l, c->ext_fr_total, c->ext_open_total);
	c->tpool_last_used = used_sum;
	dev_free(d_order); dev_free(d_k); dev_free(d_acc); if(d_open) dev_free(d_open);
	if(acc){ const double seen = 1.3 * (double)c->tpool_last_used / (double)acc, keep = c->ext_use_ratio * 0.9; c->ext_use_ratio = seen > keep ? seen : keep; if(c->ext_use_ratio < 0.2) c->ext_use_ratio = 0.2; if(c->ext_use_ratio > 1.0) c->ext_use_ratio = 1.0; }
	return WTZ_OK;
}
#endif

/* K-sw3 jobs of a batch: one wavefront per job (wtz_sw_wave.h).  WTZ_SW_SCALAR=1 forces the scalar body,
 * WTZ_SW_CHECK=1 runs both and fails loudly on any difference (on-device cross-check). */
static int run_extjobs(wtz_ctx *c, const wtz_env_t &V, wtz_extjob_t *d_jobs, uint32_t m, bool leftover = false){
	if(m == 0) return WTZ_OK;
#ifndef WTZ_EMUL
	if(leftover){
		/* behind the fused launch (run_stitch_fused): what is still open is outside the frame kernel's envelope - the general kernel takes it; every other wavefront leaves at once */
		if(!c->fused_ran) leftover = false;
		else {
			wtz_timer te; te.start();
			hipLaunchKernelGGL((wtz_kernel_extjobs<2048, 1032>), dim3(m), dim3(64), 0, g_stream, d_jobs, (const uint32_t*)NULL, m, V.P, V.pool, V.pool + 1);
			HIPCHK(hipGetLastError());
			c->cnt.ms_ext += te.stop();
			return tpool_check(c, "K-sw3 extension jobs");
		}
	}
#endif
#ifdef WTZ_EMUL
	return wtz_launch_wave<K_extjob_scalar>(0, m, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_extjob_scalar((uint32_t)t, V, d_jobs); });
#else
	const int mode = c->env_sw_mode;
	if(mode == 1) return wtz_launch_wave<K_extjob_scalar>(0, m, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_extjob_scalar((uint32_t)t, V, d_jobs); });
	std::vector<wtz_extjob_t> ref;
	if(mode == 2){
		wtz_extjob_t *d_copy = NULL; CHK(dev_alloc((void**)&d_copy, (size_t)m * sizeof(wtz_extjob_t)));
		HIPCHK(hipMemcpyAsync(d_copy, d_jobs, (size_t)m * sizeof(wtz_extjob_t), hipMemcpyDeviceToDevice, g_stream));
		CHK(wtz_launch_wave<K_extjob_scalar>(0, m, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_extjob_scalar((uint32_t)t, V, d_copy); }));
		CHK(dev_sync());
		ref.resize(m); CHK(dev_d2h(ref.data(), d_copy, (size_t)m * sizeof(wtz_extjob_t)));
		dev_free(d_copy);
	}
	/* longest-processing-time-first: the rows of an extension are sequential, so the longest job bounds the launch;
	 * start the long ones first (key = query-side length, the row count upper bound) */
	uint32_t *d_order = NULL; unsigned long long ext_sum = 0; int32_t ext_max = 0;
	std::vector<uint32_t> ord(m); std::vector<uint64_t> need(m); std::vector<uint8_t> cw(m, 0);
	unsigned long long geo_n[9] = {0}, geo_rows[9] = {0}, geo_cells[9] = {0};
	{
		/* (qlen, tlen, init_score, W) of every job: the order key, and the job's geometry = an upper bound of its trace bytes */
		int32_t *d_key = NULL; CHK(dev_alloc((void**)&d_key, (size_t)m * 16));
		CHK(wtz_launch<K_misc>(0, m, [=] WTZ_LAMBDA (uint64_t t){ const wtz_extjob_t &j = d_jobs[t]; d_key[4 * t] = j.valid ? j.qlen : -1; d_key[4 * t + 1] = j.tlen; d_key[4 * t + 2] = j.init_score; d_key[4 * t + 3] = j.W; }));
		std::vector<int32_t> key4((size_t)m * 4); CHK(dev_d2h(key4.data(), d_key, (size_t)m * 16)); dev_free(d_key);
		std::vector<int32_t> key(m);
		for(uint32_t i = 0; i < m; i++){ key[i] = key4[(size_t)i * 4]; if(key[i] > 0){ ext_sum += (unsigned long long)key[i]; if(key[i] > ext_max) ext_max = key[i]; } }
		std::vector<int32_t> rows(m, -1);       /* the order key: the rows the job can run at most (the shorter side + W, not the query side alone: most long overhangs face a short one) */
		for(uint32_t i = 0; i < m; i++){
			const int32_t qlen = key4[(size_t)i * 4], tlen = key4[(size_t)i * 4 + 1];
			need[i] = 0;
			if(qlen <= 0 || tlen <= 0) continue;
			int32_t ql = 0, n_col = 0;
			need[i] = ext_trace_need(c, qlen, tlen, key4[(size_t)i * 4 + 2], key4[(size_t)i * 4 + 3], &ql, &n_col);
			cw[i] = (uint8_t)((n_col + 63) / 64 > 255 ? 255 : (n_col + 63) / 64);
			rows[i] = ql;
			if(c->env_profile){ const int b = (n_col + 63) / 64 > 32 ? 8 : ((n_col + 63) / 64 - 1) / 4; geo_n[b]++; geo_rows[b] += (unsigned long long)ql; geo_cells[b] += (unsigned long long)ql * (unsigned long long)n_col; }
		}
		for(uint32_t i = 0; i < m; i++) ord[i] = i;
		std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b){ return rows[a] > rows[b]; });
		CHK(dev_alloc((void**)&d_order, (size_t)m * 4)); CHK(dev_h2d(d_order, ord.data(), (size_t)m * 4));
		if(c->env_profile){
			fprintf(stderr, "[ext-profile] geometry by columns per lane (upper bounds):");
			for(int b = 0; b < 9; b++) if(geo_n[b]) fprintf(stderr, " C<=%d: %llu jobs %.1f Mrows %.1f Gcells;", b < 8 ? 4 * b + 4 : 999, geo_n[b], (double)geo_rows[b] / 1e6, (double)geo_cells[b] / 1e9);
			fprintf(stderr, "\n");
		}
	}
	{
		wtz_timer te; te.start();
		const int use_reg = c->env_use_reg;
		/* launch groups: consecutive jobs of the order whose trace upper bounds fit the transient pool together; the pool is reset
		 * between groups (the CIGARs went to the main pool).  One group is the normal case. */
		const uint64_t budget = (c->pool_bytes - c->main_bytes) / 16 * 15;
		uint32_t g0 = 0, n_groups = 0;
		while(g0 < m){
			uint32_t g1 = g0; uint64_t acc = 0;
			while(g1 < m && (g1 == g0 || acc + need[ord[g1]] <= budget)){ acc += need[ord[g1]]; g1++; }
			CHK(tpool_reset(c));      /* the traces of the previous group / the previous stage are dead: their CIGARs are in the main pool */
			if(use_reg){
				/* one wavefront per job, longest first: the frame form (wtz_sw_frame.h), or the round-4 register form it replaced (WTZ_EXT_FR=0: kept as DP form 1, the
				 * reference the isolated bench compares against).  Retired in round 6 (git history keeps them): the launches per band class (WTZ_EXT_SPLIT, WTZ_EXT_FR_SPLIT:
				 * profiles/r06_ksw3_split_in_step_kernel_trace.txt) and the round-4 four-wave kernel with its WTZ_SW_MW_MIN / WTZ_EXT_MW_CW routing. */
				if(c->env_ext_fr && c->env_ext_pk){      /* two 16-bit cells per register (wtz_sw_frame16.h); what is outside its window stays open for the 32-bit form */
					hipLaunchKernelGGL((wtz_kernel_extjobs_pk<1032>), dim3(g1 - g0), dim3(64), 0, g_stream, d_jobs, (const uint32_t*)d_order + g0, g1 - g0, V.P, V.pool, V.pool + 1);
					HIPCHK(hipGetLastError());
				}
				if(c->env_ext_fr) hipLaunchKernelGGL((wtz_kernel_extjobs_fr<1032>), dim3(g1 - g0), dim3(64), 0, g_stream, d_jobs, (const uint32_t*)d_order + g0, g1 - g0, V.P, V.pool, V.pool + 1);
				else hipLaunchKernelGGL((wtz_kernel_extjobs_reg<1032>), dim3(g1 - g0), dim3(64), 0, g_stream, d_jobs, (const uint32_t*)d_order + g0, g1 - g0, V.P, V.pool, V.pool + 1);
				HIPCHK(hipGetLastError());
			}
			hipLaunchKernelGGL((wtz_kernel_extjobs<2048, 1032>), dim3(g1 - g0), dim3(64), 0, g_stream, d_jobs, (const uint32_t*)d_order + g0, g1 - g0, V.P, V.pool, V.pool + 1);     /* whatever the register DP left */
			HIPCHK(hipGetLastError());
			if(g1 < m || n_groups){ CHK(dev_sync()); CHK(tpool_check(c, "K-sw3 extension jobs")); }
			g0 = g1; n_groups++;
		}
		const double ms_l = te.stop();
		CHK(tpool_check(c, "K-sw3 extension jobs"));
		c->cnt.ms_ext += ms_l; c->cnt.n_extjobs += m;
		if(c->env_profile){
			std::vector<int32_t> key(m); uint32_t nv = 0, n256 = 0, n512 = 0, n1k = 0, n2k = 0, n4k = 0; unsigned long long s512 = 0;
			CHK(dev_sync());
			{ std::vector<wtz_extjob_t> jj(m); CHK(dev_d2h(jj.data(), d_jobs, (size_t)m * sizeof(wtz_extjob_t))); uint32_t nd[4] = {0, 0, 0, 0}; for(uint32_t i = 0; i < m; i++){ key[i] = jj[i].valid ? jj[i].x.qe : -1; if(jj[i].valid) nd[jj[i].done & 3]++; }
			  if(const char *dp = getenv("WTZ_EXT_DUMP")){      /* job geometry of this call, 8 int32 per valid job: the input of tools/ubench/ksw3_bench.py */
				if(FILE *df = fopen(dp, "ab")){ for(uint32_t i = 0; i < m; i++) if(jj[i].valid){ const int32_t r[8] = {jj[i].qlen, jj[i].tlen, jj[i].init_score, jj[i].W, jj[i].x.qe, jj[i].x.te, (int32_t)(jj[i].cells > 0x7FFFFFFFull ? 0x7FFFFFFF : jj[i].cells), (int32_t)jj[i].done}; fwrite(r, 4, 8, df); } fclose(df); } }
			  fprintf(stderr, "[ext-profile] %u launch group(s); valid jobs finished by: nobody %u, one-wave %u, four-wave %u, general %u\n", n_groups, nd[0], nd[1], nd[2], nd[3]); }
			for(uint32_t i = 0; i < m; i++){ if(key[i] < 0) continue; nv++; if(key[i] >= 256) n256++; if(key[i] >= 512){ n512++; s512 += key[i]; } if(key[i] >= 1024) n1k++; if(key[i] >= 2048) n2k++; if(key[i] >= 4096) n4k++; }
			fprintf(stderr, "[ext-profile] %u jobs (%u valid), rows (upper bound) sum %llu max %d, %.2f ms; qe>=256 %u >=512 %u (sum %llu) >=1k %u >=2k %u >=4k %u; transient pool peak %.2f GB\n", m, nv, ext_sum, ext_max, ms_l, n256, n512, s512, n1k, n2k, n4k, c->tpool_peak_call / 1073741824.0);
		}
		dev_free(d_order);
	}
	if(mode == 2){
		CHK(dev_sync());
		std::vector<wtz_extjob_t> got(m); CHK(dev_d2h(got.data(), d_jobs, (size_t)m * sizeof(wtz_extjob_t)));
		for(uint32_t i = 0; i < m; i++){
			if(!got[i].valid) continue;
			if(memcmp(&got[i].x, &ref[i].x, sizeof(wtz_aln_t)) || got[i].cigar_len != ref[i].cigar_len || got[i].cells != ref[i].cells)
				return wtz_fail(WTZ_E_STATE, "K-sw3 wave kernel differs from the scalar body on job %u: qlen %d tlen %d init %d W %d; score %d/%d qe %d/%d te %d/%d aln %d/%d cigar %u/%u cells %llu/%llu",
					i, got[i].qlen, got[i].tlen, got[i].init_score, got[i].W, got[i].x.score, ref[i].x.score, got[i].x.qe, ref[i].x.qe, got[i].x.te, ref[i].x.te,
					got[i].x.aln, ref[i].x.aln, got[i].cigar_len, ref[i].cigar_len, (unsigned long long)got[i].cells, (unsigned long long)ref[i].cells);
			if(got[i].cigar_len){
				std::vector<uint32_t> a(got[i].cigar_len), b(got[i].cigar_len);
				CHK(dev_d2h(a.data(), got[i].cigar, a.size() * 4)); CHK(dev_d2h(b.data(), ref[i].cigar, b.size() * 4));
				if(a != b) return wtz_fail(WTZ_E_STATE, "K-sw3 wave kernel: CIGAR differs on job %u", i);
			}
		}
	}
	return WTZ_OK;
#endif
}


/* ------------------------------------------------------------------------------------------------ */
/* A9 with one lane per K-sw1 problem (wtz_sw_lane.h)                                                */
/* ------------------------------------------------------------------------------------------------ */
#ifdef WTZ_EMUL
#define WTZ_HOST_WAVE 1u
#else
#define WTZ_HOST_WAVE 64u
#endif
/* a block of the main (0) / transient (1) device pool for a host-side array */
static int pool_alloc_host(wtz_ctx *c, int which, size_t bytes, void **out){
	unsigned long long *d_p = NULL; CHK(dev_alloc((void**)&d_p, 8));
	wtz_pool_t *pool = c->dpool + which;
	CHK(wtz_launch<K_poolalloc>(0, 1, [=] WTZ_LAMBDA (uint64_t){ *d_p = (unsigned long long)(uintptr_t)wtz_pool_alloc(pool, bytes); }));
	unsigned long long h = 0; CHK(dev_d2h(&h, d_p, 8)); dev_free(d_p);
	if(h == 0) return wtz_fail(WTZ_E_POOL, "device scratch pool exhausted (%zu bytes for the K-sw1 problem lists)", bytes);
	*out = (void*)(uintptr_t)h; return WTZ_OK;
}
/* windows d_wt[0, nwt): plan -> shape sort -> relative-mode DP per class -> fold.  d_fb ([0] = count, room for nwt + 1) receives the
 * windows the chained kernel has to do (outside the envelope, or an absolute test of kswx_extend_align_core would have fired). */
static int run_winalign_lane(wtz_ctx *c, const wtz_env_t &V, const wtz_wintask_t *d_wt, uint64_t nwt, const wtz_alnitem_t *d_items, uint32_t *d_fb, uint32_t *n_fb){
	*n_fb = 0;
	if(nwt == 0) return WTZ_OK;
	const bool prof = c->env_profile; double tp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp0 = 0;
	if(prof){ (void)dev_sync(); tp0 = wtz_wall(); }
	auto lap = [&](int k){ if(prof){ (void)dev_sync(); const double t = wtz_wall(); tp[k] += t - tp0; tp0 = t; } };
	uint32_t *d_na = NULL, *d_woff = NULL;
	CHK(dev_alloc((void**)&d_na, (nwt + 1) * 4)); CHK(dev_alloc((void**)&d_woff, (nwt + 1) * 4));
	CHK(dev_set(d_na, 0, (nwt + 1) * 4));
	CHK(wtz_launch<K_lcount>(0, nwt, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_lcount((uint32_t)t, d_wt, d_items, d_na); }));
	CHK(dev_exclusive_scan_u32(d_na, d_woff, nwt + 1));
	uint32_t NS = 0; CHK(dev_d2h(&NS, d_woff + nwt, 4));
	lap(0);
	wtz_lprob_t *d_prob = NULL; uint32_t *d_cap = NULL, *d_roff = NULL, *d_val = NULL, *d_ccnt = NULL, *d_uidx = NULL, *d_nu = NULL; uint64_t *d_key = NULL; uint8_t *d_flag = NULL; wtz_lres_t *d_res = NULL;
	{   /* per-slot arrays: one block of the main pool */
		const size_t nsp = (size_t)NS + 64;
		const size_t b_prob = (nsp * sizeof(wtz_lprob_t) + 255) & ~(size_t)255, b_res = (nsp * sizeof(wtz_lres_t) + 255) & ~(size_t)255, b_u32 = (nsp * 4 + 255) & ~(size_t)255, b_u64 = (nsp * 8 + 255) & ~(size_t)255;
		uint8_t *blk = NULL; CHK(pool_alloc_host(c, 0, b_prob + b_res + 4 * b_u32 + b_u64, (void**)&blk));
		d_prob = (wtz_lprob_t*)blk; blk += b_prob; d_res = (wtz_lres_t*)blk; blk += b_res; d_cap = (uint32_t*)blk; blk += b_u32; d_roff = (uint32_t*)blk; blk += b_u32; d_val = (uint32_t*)blk; blk += b_u32; d_uidx = (uint32_t*)blk; blk += b_u32; d_key = (uint64_t*)blk;
	}
	CHK(dev_alloc((void**)&d_flag, nwt + 16)); CHK(dev_alloc((void**)&d_nu, (nwt + 1) * 4)); CHK(dev_alloc((void**)&d_ccnt, 32)); CHK(dev_set(d_ccnt, 0, 32));
	CHK(dev_set(d_cap, 0, ((size_t)NS + 1) * 4));
	CHK(dev_set(d_res, 0, ((size_t)NS + 1) * sizeof(wtz_lres_t)));
	CHK(wtz_launch<K_lplan>(0, nwt, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_lplan((uint32_t)t, V, d_wt, d_items, d_woff, d_prob, d_cap, d_key, d_val, d_flag, d_ccnt, d_uidx, d_nu); }));
	lap(1);
	CHK(dev_exclusive_scan_u32(d_cap, d_roff, (uint64_t)NS + 1));
	uint32_t NR = 0, ccnt[4] = {0, 0, 0, 0};
	CHK(dev_d2h(&NR, d_roff + NS, 4)); CHK(dev_d2h(ccnt, d_ccnt, 16));
	uint32_t *d_runs = NULL; CHK(pool_alloc_host(c, 0, ((size_t)NR + 16) * 4, (void**)&d_runs));
	lap(2);
	CHK(dev_sort_pairs_u64_u32(d_key, d_val, NS, 16));
	lap(3);                 /* ascending inverted key = widest band first, longest first inside a width */
	CHK(dev_set(d_fb, 0, 4));
	{
		const uint32_t *d_ord = d_val; const wtz_lprob_t *pp = d_prob; const uint32_t *ro = d_roff; uint32_t *rn = d_runs; wtz_lres_t *rs = d_res;
		wtz_lclass_t L; uint32_t lo = 0, wv = 0;
		for(int k = 0; k < 4; k++){ const uint32_t n = ccnt[3 - k]; L.lo[k] = lo; L.hi[k] = lo + n; wv += (n + WTZ_HOST_WAVE - 1) / WTZ_HOST_WAVE; L.wend[k] = wv; lo += n; }
		uint64_t *d_wtr = NULL; uint32_t *d_wrm = NULL;
		CHK(dev_alloc((void**)&d_wtr, ((size_t)wv + 1) * 8)); CHK(dev_alloc((void**)&d_wrm, ((size_t)wv + 1) * 4)); CHK(dev_set(d_wtr, 0, ((size_t)wv + 1) * 8));
		if(wv) CHK(wtz_launch_coop<K_ldp>(0, wv, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_ldp_all((uint32_t)t, L, V, d_wt, d_items, d_ord, pp, rs, d_wtr, d_wrm); }, 0));
		lap(4);
		if(wv) CHK(wtz_launch_coop<K_ltb>(0, wv, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_ltb_all((uint32_t)t, L, V, d_wt, d_items, d_ord, pp, ro, rn, rs, d_wtr, d_wrm); }, 0));
		lap(6);
		CHK(wtz_launch<K_lfold>(0, nwt, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_lfold((uint32_t)t, V, d_wt, d_items, d_woff, pp, ro, rn, rs, d_flag, d_fb, d_uidx, d_nu); }));
	}
	CHK(dev_d2h(n_fb, d_fb, 4));
	if(*n_fb > 64){
		/* the windows left to the chained wave kernel, the ones with the most anchors first (a window's chain of problems is sequential; K_lfold appends in the order its lanes
		 * arrive: the same tail as the gap list's, run_gap_lane) */
		const uint32_t nl = *n_fb; uint64_t *d_k2 = NULL; CHK(dev_alloc((void**)&d_k2, (size_t)nl * 8));
		const uint32_t *lst = d_fb + 1;
		CHK(wtz_launch<K_misc>(0, nl, [=] WTZ_LAMBDA (uint64_t i){
			const wtz_wintask_t tk = d_wt[lst[i]];
			const wtz_win_t &w = d_items[tk.item].win[tk.widx];
			d_k2[i] = 0xFFFFFFFFull - (unsigned long long)(w.anchors[1] - w.anchors[0]);
		}));
		CHK(dev_sort_pairs_u64_u32(d_k2, d_fb + 1, nl, 32));
		dev_free(d_k2);
	}
	lap(5);
	if(c->env_profile) fprintf(stderr, "[lane-profile] %llu windows, %u anchor slots, K-sw1 problems by band class <=16 / <=32 / <=64 / <=104: %u / %u / %u / %u, %u run entries, %u windows left to the chained kernel; ms: count+scan %.2f plan %.2f scan+alloc %.2f sort %.2f dp %.2f traceback %.2f fold %.2f\n",
		(unsigned long long)nwt, NS, ccnt[0], ccnt[1], ccnt[2], ccnt[3], NR, *n_fb, tp[0] * 1e3, tp[1] * 1e3, tp[2] * 1e3, tp[3] * 1e3, tp[4] * 1e3, tp[6] * 1e3, tp[5] * 1e3);
	dev_free(d_na); dev_free(d_woff); dev_free(d_flag); dev_free(d_ccnt);
	return WTZ_OK;
}


/* K-sw2 gaps of the window slots d_wt[0, nwt) with one lane per gap (wtz_lane_global); d_list ([0] = count, room for nwt + 1) = the slots
 * the wavefront kernel still has to do (empty sides, bands beyond 104 columns, gaps whose band has to be doubled again) */
static int run_gap_lane(wtz_ctx *c, const wtz_env_t &V, const wtz_wintask_t *d_wt, uint64_t nwt, const wtz_alnitem_t *d_items, wtz_gapres_t *d_gaps, uint32_t *d_list, uint32_t *n_list){
	*n_list = 0;
	if(nwt == 0) return WTZ_OK;
	wtz_lgap_t *d_gp = NULL; uint32_t *d_cap = NULL, *d_roff = NULL, *d_val = NULL, *d_ccnt = NULL; uint64_t *d_key = NULL; uint8_t *d_done = NULL;
	{
		const size_t nsp = (size_t)nwt + 64;
		const size_t b_gp = (nsp * sizeof(wtz_lgap_t) + 255) & ~(size_t)255, b_u32 = (nsp * 4 + 255) & ~(size_t)255, b_u64 = (nsp * 8 + 255) & ~(size_t)255, b_u8 = (nsp + 255) & ~(size_t)255;
		uint8_t *blk = NULL; CHK(pool_alloc_host(c, 0, b_gp + 3 * b_u32 + b_u64 + b_u8, (void**)&blk));
		d_gp = (wtz_lgap_t*)blk; blk += b_gp; d_cap = (uint32_t*)blk; blk += b_u32; d_roff = (uint32_t*)blk; blk += b_u32; d_val = (uint32_t*)blk; blk += b_u32; d_key = (uint64_t*)blk; blk += b_u64; d_done = blk;
	}
	CHK(dev_alloc((void**)&d_ccnt, 32)); CHK(dev_set(d_ccnt, 0, 32));
	CHK(dev_set(d_cap, 0, ((size_t)nwt + 1) * 4));
	CHK(wtz_launch<K_gplan>(0, nwt, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_gplan((uint32_t)t, V, d_wt, d_items, d_gaps, d_gp, d_cap, d_key, d_val, d_done, d_ccnt); }));
	CHK(dev_exclusive_scan_u32(d_cap, d_roff, nwt + 1));
	uint32_t NR = 0, ccnt[4] = {0, 0, 0, 0};
	CHK(dev_d2h(&NR, d_roff + nwt, 4)); CHK(dev_d2h(ccnt, d_ccnt, 16));
	uint32_t *d_runs = NULL; CHK(pool_alloc_host(c, 0, ((size_t)NR + 16) * 4, (void**)&d_runs));
	CHK(dev_sort_pairs_u64_u32(d_key, d_val, nwt, 18));
	CHK(dev_set(d_list, 0, 4));
	{
		const uint32_t *d_ord = d_val; const wtz_lgap_t *gp = d_gp; const uint32_t *ro = d_roff; uint32_t *rn = d_runs; uint8_t *dn = d_done;
		wtz_lclass_t L; uint32_t lo = 0, wv = 0;
		for(int k = 0; k < 4; k++){ const uint32_t n = ccnt[3 - k]; L.lo[k] = lo; L.hi[k] = lo + n; wv += (n + WTZ_HOST_WAVE - 1) / WTZ_HOST_WAVE; L.wend[k] = wv; lo += n; }
		uint64_t *d_wtr = NULL; uint32_t *d_wrm = NULL; wtz_lres_t *d_res = NULL;
		CHK(dev_alloc((void**)&d_wtr, ((size_t)wv + 1) * 8)); CHK(dev_alloc((void**)&d_wrm, ((size_t)wv + 1) * 4)); CHK(dev_set(d_wtr, 0, ((size_t)wv + 1) * 8));
		CHK(pool_alloc_host(c, 0, ((size_t)nwt + 1) * sizeof(wtz_lres_t), (void**)&d_res)); CHK(dev_set(d_res, 0, ((size_t)nwt + 1) * sizeof(wtz_lres_t)));
		if(wv) CHK(wtz_launch_coop<K_gdp>(0, wv, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_gdp_all((uint32_t)t, L, V, d_wt, d_items, d_ord, gp, d_res, d_wtr, d_wrm); }, 0));
		if(wv) CHK(wtz_launch_coop<K_gtb>(0, wv, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_gtb_all((uint32_t)t, L, V, d_wt, d_items, d_ord, gp, ro, rn, d_res, d_gaps, dn, d_wtr, d_wrm); }, 0));
		CHK(wtz_launch<K_glist>(0, nwt, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_glist((uint32_t)t, dn, d_list); }));
	}
	CHK(dev_d2h(n_list, d_list, 4));
	if(*n_list > 64){
		/* the gaps left to the wavefront kernel, heaviest first (rows x lane-columns of the first band): K_glist appends them in whatever order its lanes arrive, and a
		 * launch of a few thousand wave-sized DPs of very different lengths ended in the tail of whichever long one started last (round 6) */
		const uint32_t nl = *n_list; uint64_t *d_k2 = NULL; CHK(dev_alloc((void**)&d_k2, (size_t)nl * 8));
		const wtz_lgap_t *gp = d_gp; const uint32_t *lst = d_list + 1;
		CHK(wtz_launch<K_misc>(0, nl, [=] WTZ_LAMBDA (uint64_t i){
			const wtz_lgap_t G = gp[lst[i]];
			const int32_t nc = G.dq < 2 * G.w + 1 ? G.dq : 2 * G.w + 1;
			unsigned long long wgt = (unsigned long long)(G.dt > 0 ? G.dt : 0) * (unsigned long long)((nc > 0 ? nc : 0) / 64 + 1);
			if(wgt > 0xFFFFFFFEull) wgt = 0xFFFFFFFEull;
			d_k2[i] = 0xFFFFFFFFull - wgt;
		}));
		CHK(dev_sort_pairs_u64_u32(d_k2, d_list + 1, nl, 32));
		dev_free(d_k2);
	}
	if(c->env_profile) fprintf(stderr, "[lane-profile] %llu window slots, K-sw2 gaps by band class <=16 / <=32 / <=64 / <=104: %u / %u / %u / %u, %u left to the wavefront kernel\n",
		(unsigned long long)nwt, ccnt[0], ccnt[1], ccnt[2], ccnt[3], *n_list);
	dev_free(d_ccnt);
	return WTZ_OK;
}

extern "C" int wtz_pairs_align(wtz_ctx_t *c, const uint32_t *pair_idx, const uint8_t *dir, uint32_t m, wtz_aln_result_t *out){
	if(!c || !c->have_pairs) return wtz_fail(WTZ_E_STATE, "wtz_pairs_align before wtz_pairs_seed");
	if(m == 0) return WTZ_OK;
	CTX_ENTER(c);
	if(!pair_idx || !dir || !out) return wtz_fail(WTZ_E_ARG, "null argument");
	c->n_items = 0; c->have_items = false;
	const bool prof_wall = c->env_profile; double tw[6] = {0, 0, 0, 0, 0, 0}; double tw0 = prof_wall ? wtz_wall() : 0;
	auto lapw = [&](int k){ if(prof_wall){ const double t = wtz_wall(); tw[k] += t - tw0; tw0 = t; } };
	CHK(reserve_items(c, m));
	std::vector<wtz_alnitem_t> items(m); uint64_t nreg = 0;      /* the window tasks (item, window) are listed on the device: one per region slot, in slot order */
	std::vector<uint32_t> h_q(c->n_pairs), h_c(c->n_pairs);
	CHK(dev_d2h(h_q.data(), c->d_qid, (size_t)c->n_pairs * 4)); CHK(dev_d2h(h_c.data(), c->d_cid, (size_t)c->n_pairs * 4));
	for(uint32_t i = 0; i < m; i++){
		if(pair_idx[i] >= c->n_pairs || dir[i] > 1) return wtz_fail(WTZ_E_ARG, "align item %u out of range", i);
		const wtz_pairres_t &r = c->h_pairres[pair_idx[i]];
		wtz_alnitem_t it; it.q = h_q[pair_idx[i]]; it.c = h_c[pair_idx[i]]; it.dir = dir[i];
		it.win = r.win[dir[i]]; it.anchors = r.anchors[dir[i]]; it.nwin = r.nwin[dir[i]]; it.regs = (wtz_reg_t*)(uintptr_t)nreg;
		nreg += it.nwin; items[i] = it;
	}
	wtz_reg_t *d_regs = NULL, *d_regs_chk = NULL; wtz_alnitem_t *d_items = NULL; wtz_wintask_t *d_wt = NULL;
	CHK(dev_alloc((void**)&d_regs, (size_t)(nreg + 1) * sizeof(wtz_reg_t)));
	for(uint32_t i = 0; i < m; i++) items[i].regs = d_regs + (uintptr_t)items[i].regs;
	CHK(dev_alloc((void**)&d_items, (size_t)m * sizeof(wtz_alnitem_t))); CHK(dev_h2d(d_items, items.data(), (size_t)m * sizeof(wtz_alnitem_t)));
	CHK(dev_alloc((void**)&d_wt, (size_t)(nreg + 1) * sizeof(wtz_wintask_t)));
	{ const wtz_alnitem_t *di = d_items; wtz_wintask_t *dw = d_wt; const wtz_reg_t *r0 = d_regs;
	  CHK(wtz_launch<K_misc>(0, m, [=] WTZ_LAMBDA (uint64_t i){ const wtz_alnitem_t &it = di[i]; wtz_wintask_t *w = dw + (it.regs - r0); for(uint32_t k = 0; k < it.nwin; k++){ w[k].item = (uint32_t)i; w[k].widx = k; } })); }
	const wtz_env_t V = ctx_env(c); wtz_alnres_dev_t *d_res = c->d_alnres;
	lapw(0);
	wtz_timer tm; tm.start();
	bool lane_done = false;
	if(c->env_lane && !c->env_grp4){
		/* one lane per K-sw1 problem (wtz_sw_lane.h); what it leaves (d_fb) goes through the chained kernel below */
		const uint64_t nwt0 = (size_t)nreg;
		uint32_t *d_fb = NULL, n_fb = 0;
		CHK(dev_alloc((void**)&d_fb, (nwt0 + 1) * 4));
		STAGE(c, "K-sw1 lane pipeline");
		CHK(run_winalign_lane(c, V, d_wt, nwt0, d_items, d_fb, &n_fb));
#ifdef WTZ_EMUL
		if(n_fb) CHK(wtz_launch_coop<K_winalign>(0, n_fb, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_winalign((uint32_t)t, V, d_wt, d_items, (uint32_t*)NULL, d_fb); }, WTZ_WINALIGN_LDS_BYTES + WTZ_WINALIGN_QW_BYTES));
#else
		if(n_fb){
			uint32_t *d_defer = NULL, n_def = 0; CHK(dev_alloc((void**)&d_defer, ((size_t)n_fb + 1) * 4)); CHK(dev_set(d_defer, 0, 4));
			CHK(wtz_launch_coop<K_winalign>(0, n_fb, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_winalign<false>((uint32_t)t, V, d_wt, d_items, d_defer, d_fb); }, WTZ_WINALIGN_LDS_BYTES + WTZ_WINALIGN_QW_BYTES));
			CHK(dev_d2h(&n_def, d_defer, 4));
			if(n_def) CHK(wtz_launch_coop<K_winalign_big>(0, n_def, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_winalign<true>((uint32_t)t, V, d_wt, d_items, NULL, d_defer); }, WTZ_WINALIGN_LDS_BYTES + WTZ_WINALIGN_QW_BYTES));
			dev_free(d_defer);
		}
#endif
		dev_free(d_fb);
		lane_done = (c->env_lane != 2);
		if(c->env_lane == 2){ CHK(dev_sync()); CHK(dev_alloc((void**)&d_regs_chk, (size_t)(nreg + 1) * sizeof(wtz_reg_t))); CHK(dev_d2d(d_regs_chk, d_regs, (size_t)nreg * sizeof(wtz_reg_t))); }
	}
#ifdef WTZ_EMUL
	if(!lane_done) CHK(wtz_launch_coop<K_winalign>(0, (size_t)nreg, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_winalign((uint32_t)t, V, d_wt, d_items); }, WTZ_WINALIGN_LDS_BYTES + WTZ_WINALIGN_QW_BYTES));
#else
	if(!lane_done){
		/* first launch: FOUR windows per wavefront (one per 16-lane group, wtz_sw_grp.h); a window with a problem outside the group form's
		 * envelope queues itself for the one-window-per-wave kernel: its lean form first (register DP with one / two band columns per lane,
		 * no scalar body: fewer VGPRs), then - for what that form defers in turn - the full task */
		const uint64_t nwt0 = (size_t)nreg;
		uint32_t *d_defer4 = NULL, *d_defer = NULL;
		CHK(dev_alloc((void**)&d_defer4, (nwt0 + 1) * 4)); CHK(dev_set(d_defer4, 0, 4));
		CHK(dev_alloc((void**)&d_defer, (nwt0 + 1) * 4)); CHK(dev_set(d_defer, 0, 4));
		STAGE(c, "K_winalign");
		uint32_t n_def4 = 0, n_def = 0;
		if(c->env_grp4){
			CHK(wtz_launch_grp<K_winalign4>(nwt0, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_winalign4((uint32_t)t, (uint32_t)nwt0, V, d_wt, d_items, d_defer4); }, WTZ_WINALIGN4_LDS_BYTES));
			CHK(dev_d2h(&n_def4, d_defer4, 4));
			if(n_def4) CHK(wtz_launch_coop<K_winalign>(0, n_def4, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_winalign<false>((uint32_t)t, V, d_wt, d_items, d_defer, d_defer4); }, WTZ_WINALIGN_LDS_BYTES + WTZ_WINALIGN_QW_BYTES));
		} else {
			CHK(wtz_launch_coop<K_winalign>(0, nwt0, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_winalign<false>((uint32_t)t, V, d_wt, d_items, d_defer, NULL); }, WTZ_WINALIGN_LDS_BYTES + WTZ_WINALIGN_QW_BYTES));
		}
		CHK(dev_d2h(&n_def, d_defer, 4));
		if(n_def) CHK(wtz_launch_coop<K_winalign_big>(0, n_def, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_winalign<true>((uint32_t)t, V, d_wt, d_items, NULL, d_defer); }, WTZ_WINALIGN_LDS_BYTES + WTZ_WINALIGN_QW_BYTES));
		if(c->env_profile) fprintf(stderr, "[winalign-profile] %zu windows, %u left by the four-per-wave form, %u redone by the full task\n", (size_t)nreg, n_def4, n_def);
		dev_free(d_defer4);
		dev_free(d_defer);
	}
#endif
	CHK(dev_sync());
	if(d_regs_chk){
		/* WTZ_WINALIGN_LANE=2: every window was aligned by the lane pipeline AND by the chained kernel: any difference is fatal */
		unsigned int *d_bad = NULL; CHK(dev_alloc((void**)&d_bad, 16)); CHK(dev_set(d_bad, 0, 16));
		const wtz_reg_t *ra = d_regs, *rb = d_regs_chk;
		CHK(wtz_launch<K_misc>(0, nreg, [=] WTZ_LAMBDA (uint64_t i){
			const wtz_reg_t &a = ra[i], &b = rb[i];
			bool same = a.pass == 2 || b.pass == 2 || (a.x.score == b.x.score && a.x.tb == b.x.tb && a.x.te == b.x.te && a.x.qb == b.x.qb && a.x.qe == b.x.qe && a.x.aln == b.x.aln && a.x.mat == b.x.mat && a.x.mis == b.x.mis && a.x.ins == b.x.ins && a.x.del == b.x.del && a.pass == b.pass && a.cigar_len == b.cigar_len && (a.cells == b.cells || a.cells == 0 || b.cells == 0));      /* pass 2 = scratch exhausted (reported as such); the host emulation's scalar body does not count cells */
			if(same && a.pass != 2 && b.pass != 2) for(uint32_t k = 0; k < a.cigar_len; k++) if(a.cigar[k] != b.cigar[k]){ same = false; break; }
			if(!same){ const unsigned int z = WTZ_ATOMIC_INC32(&d_bad[0]); if(z == 0) d_bad[1] = (unsigned int)i; }
		}));
		unsigned int hb[4]; CHK(dev_d2h(hb, d_bad, 16)); dev_free(d_bad);
		if(hb[0]){
			wtz_reg_t a, b; CHK(dev_d2h(&a, d_regs + hb[1], sizeof a)); CHK(dev_d2h(&b, d_regs_chk + hb[1], sizeof b));
			return wtz_fail(WTZ_E_STATE, "K-sw1 lane pipeline differs from the chained kernel on %u of %llu windows; first: window slot %u chained/lane score %d/%d tb %d/%d te %d/%d qb %d/%d qe %d/%d aln %d/%d mat %d/%d mis %d/%d ins %d/%d del %d/%d cigar %u/%u pass %u/%u cells %llu/%llu",
				hb[0], (unsigned long long)nreg, hb[1], a.x.score, b.x.score, a.x.tb, b.x.tb, a.x.te, b.x.te, a.x.qb, b.x.qb, a.x.qe, b.x.qe, a.x.aln, b.x.aln, a.x.mat, b.x.mat, a.x.mis, b.x.mis, a.x.ins, b.x.ins, a.x.del, b.x.del, a.cigar_len, b.cigar_len, a.pass, b.pass, a.cells, b.cells);
		}
		dev_free(d_regs_chk);
	}
	c->cnt.ms_winalign += tm.stop(); c->cnt.n_winalign += (size_t)nreg;
	lapw(1);
	tm.start();
	{
		wtz_stitch_state_t *d_st = NULL; wtz_extjob_t *d_jl = NULL, *d_jr = NULL;
		CHK(dev_alloc((void**)&d_st, (size_t)m * sizeof(wtz_stitch_state_t)));
		CHK(dev_alloc((void**)&d_jl, (size_t)m * sizeof(wtz_extjob_t))); CHK(dev_alloc((void**)&d_jr, (size_t)m * sizeof(wtz_extjob_t)));
		wtz_gapres_t *d_gaps = NULL; CHK(dev_alloc((void**)&d_gaps, (size_t)(nreg + 1) * sizeof(wtz_gapres_t)));
		const uint64_t nwt = (size_t)nreg;
		STAGE(c, "K_stitch_left");
		int32_t *d_rgeo = NULL;
#ifndef WTZ_EMUL
		const bool fused = c->env_ext_fused && c->env_ext_fr && c->env_sw_mode == 0 && c->env_use_reg;
		if(fused) CHK(dev_alloc((void**)&d_rgeo, (size_t)m * 8));
#else
		CHK(dev_alloc((void**)&d_rgeo, (size_t)m * 8));      /* host emulation: the prediction of the right extension's geometry is compared with what K_stitch_mid asks for */
#endif
		CHK(wtz_launch_wave<K_stitch_left>(0, m, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_stitch_left((uint32_t)t, V, d_items, d_st, d_jl, d_rgeo); }));
		uint32_t *d_glist = NULL, n_glist = (uint32_t)nwt; const bool gap_lane = c->env_gap_lane != 0;
#ifndef WTZ_EMUL
		wtz_timer tgap; tgap.start();          /* K-sw2: lane pipeline + wavefront kernel */
#endif
		if(gap_lane){
			CHK(dev_alloc((void**)&d_glist, (size_t)(nwt + 1) * 4));
			STAGE(c, "K-sw2 lane pipeline");
			CHK(run_gap_lane(c, V, d_wt, nwt, d_items, d_gaps, d_glist, &n_glist));
		}
#ifdef WTZ_EMUL
		CHK(wtz_launch_coop<K_gap>(0, n_glist, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_gap((uint32_t)t, V, d_wt, d_items, d_gaps, (uint32_t*)NULL, (const uint32_t*)d_glist); }, WTZ_GAP_LDS_BYTES));
		CHK(run_extjobs(c, V, d_jl, m));
#else
		{
			/* the gaps between windows (many short K-sw2 tasks) and the left extensions (few long K-sw3 jobs) are independent: the
			 * gap kernel can run on its own stream and fill the CUs the extension tail leaves idle; stitch_mid waits for both */
			/* measured: ~5 ms of 150 on the E. coli shape, inside run-to-run noise, and it folds K_gap's contention into the K-sw3 stage
			 * time that bench.py reports against the roofline -> opt-in (WTZ_GAP_SIDESTREAM=1) */
			const int gap_side = c->env_gap_side;
			hipStream_t main_stream = g_stream;
			if(gap_side){ HIPCHK(hipEventRecord(c->ev_gap_fork, main_stream)); HIPCHK(hipStreamWaitEvent(c->stream_gap, c->ev_gap_fork, 0)); g_stream = c->stream_gap; }
			uint32_t *d_defer = NULL; CHK(dev_alloc((void**)&d_defer, (size_t)(nwt + 1) * 4)); CHK(dev_set(d_defer, 0, 4));
			STAGE(c, "K_gap");
			int rc_gap = wtz_launch_coop<K_gap>(0, n_glist, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_gap((uint32_t)t, V, d_wt, d_items, d_gaps, d_defer, (const uint32_t*)d_glist, 0u); }, WTZ_GAP_LDS_BYTES);
			if(rc_gap == WTZ_OK){
				/* gaps whose band outgrew the register forms (repeats): the LDS-ring wave DP with 8192-column rings, 72 KB of LDS per wave */
				uint32_t n_def = 0; rc_gap = dev_d2h(&n_def, d_defer, 4);
				if(rc_gap == WTZ_OK && n_def) rc_gap = wtz_launch_coop<K_gap_wide>(0, n_def, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_gap((uint32_t)t, V, d_wt, d_items, d_gaps, NULL, d_defer, (uint32_t)WTZ_GAP_WIDE_LDS_BYTES); }, WTZ_GAP_WIDE_LDS_BYTES);
				if(rc_gap == WTZ_OK && c->env_profile){ rc_gap = dev_sync(); fprintf(stderr, "[gap-profile] %llu window slots, %u wide gaps redone with 72 KB of LDS\n", (unsigned long long)nwt, n_def); }
			}
			if(rc_gap == WTZ_OK && !gap_side){ tgap.lap(); }
			g_stream = main_stream;
			CHK(rc_gap);
			if(gap_side) HIPCHK(hipEventRecord(c->ev_gap_join, c->stream_gap));
			if(fused){
				if(gap_side) HIPCHK(hipStreamWaitEvent(main_stream, c->ev_gap_join, 0));      /* the join inside the fused launch reads the gaps */
				STAGE(c, "extjobs left + join + right on one wavefront"); CHK(run_stitch_fused(c, V, d_items, d_st, d_jl, d_jr, d_gaps, d_rgeo, m));
			}
			STAGE(c, "extjobs left");
			CHK(run_extjobs(c, V, d_jl, m, fused));
			if(gap_side) HIPCHK(hipStreamWaitEvent(main_stream, c->ev_gap_join, 0));
			else c->cnt.ms_gap += tgap.read();      /* after the extension jobs: no extra synchronisation for the lap */
		}
#endif
		STAGE(c, "K_stitch_mid");
		CHK(wtz_launch_coop<K_stitch_mid>(0, m, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_stitch_mid((uint32_t)t, V, d_items, d_st, d_jl, d_jr, d_gaps); }));
		STAGE(c, "extjobs right");
#ifdef WTZ_EMUL
		for(uint32_t t = 0; t < m; t++){
			const wtz_extjob_t &jr = d_jr[t];
			const int32_t pq = d_rgeo[2 * (size_t)t], pt = d_rgeo[2 * (size_t)t + 1];
			if(jr.valid ? (pq != jr.qlen || pt != jr.tlen) : (pq >= 0 && !d_st[t].bad))
				return wtz_fail(WTZ_E_STATE, "stitch: predicted right extension of item %u (%d x %d) differs from the one K_stitch_mid asks for (valid %u: %d x %d)", t, pq, pt, jr.valid, jr.qlen, jr.tlen);
		}
		CHK(run_extjobs(c, V, d_jr, m));
#else
		CHK(run_extjobs(c, V, d_jr, m, fused));
#endif
		STAGE(c, "K_stitch_fin");
		CHK(wtz_launch_coop<K_stitch_fin>(0, m, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_stitch_fin((uint32_t)t, V, d_items, d_st, d_jl, d_jr, d_res); }));
		if(c->P.refine) STAGE(c, "K_refine");
		if(c->P.refine) CHK(wtz_launch_coop<K_refine>(0, m, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_refine((uint32_t)t, V, d_items, d_res); }, WTZ_REFINE_LDS_BYTES));
		CHK(dev_sync());
		dev_free(d_st); dev_free(d_jl); dev_free(d_jr); dev_free(d_gaps);
	}
	c->cnt.ms_stitch += tm.stop(); c->cnt.n_stitch += m;
	lapw(2);
	c->h_alnres.resize(m); c->n_items = m; c->have_items = true;
	CHK(dev_d2h(c->h_alnres.data(), c->d_alnres, (size_t)m * sizeof(wtz_alnres_dev_t)));
	dev_free(d_regs); dev_free(d_items); dev_free(d_wt);
	CHK(pool_check(c, "wtz_pairs_align"));
	uint64_t coff = 0, toff = 0;
	for(uint32_t i = 0; i < m; i++){
		const wtz_alnres_dev_t &r = c->h_alnres[i];
		if(r.bad) return wtz_fail(WTZ_E_POOL, "wtz_pairs_align: item %u ran out of scratch", i);
		wtz_aln_result_t o; memset(&o, 0, sizeof o);
		o.score = r.x.score; o.tb = r.x.tb; o.te = r.x.te; o.qb = r.x.qb; o.qe = r.x.qe; o.aln = r.x.aln; o.mat = r.x.mat; o.mis = r.x.mis; o.ins = r.x.ins; o.del = r.x.del;
		o.n_regs = r.n_regs; o.cigar_len = r.cigar_len; o.cigar_off = coff; coff += r.cigar_len; o.text_len = r.text_len; o.text_off = toff; toff += r.text_len;
		c->cnt.cells_shift += r.cells_shift; c->cnt.cells_fixed += r.cells_fixed; c->cnt.cells_global += r.cells_global;
		out[i] = o;
	}
	lapw(3);
	if(prof_wall) fprintf(stderr, "[align-profile] %u items: host wall ms prep %.2f winalign %.2f stitch %.2f results %.2f\n", m, tw[0] * 1e3, tw[1] * 1e3, tw[2] * 1e3, tw[3] * 1e3);
	return WTZ_OK;
}

extern "C" int wtz_fetch_cigars(wtz_ctx_t *c, uint32_t *dst, uint64_t n_ops){
	if(!c || !c->have_items) return wtz_fail(WTZ_E_STATE, "wtz_fetch_cigars before wtz_pairs_align");
	CTX_ENTER(c);
	uint64_t tot = 0; for(uint32_t i = 0; i < c->n_items; i++) tot += c->h_alnres[i].cigar_len;
	if(tot != n_ops) return wtz_fail(WTZ_E_ARG, "wtz_fetch_cigars: expected room for %llu ops, got %llu", (unsigned long long)tot, (unsigned long long)n_ops);
	if(tot == 0) return WTZ_OK;
	if(!dst) return wtz_fail(WTZ_E_ARG, "null output");
	std::vector<uint64_t> off((size_t)c->n_items + 1);
	uint64_t o = 0; for(uint32_t i = 0; i < c->n_items; i++){ off[i] = o; o += c->h_alnres[i].cigar_len; } off[c->n_items] = o;
	uint64_t *d_off = NULL; uint32_t *d_c = NULL;
	CHK(dev_alloc((void**)&d_off, off.size() * 8)); CHK(dev_h2d(d_off, off.data(), off.size() * 8));
	CHK(dev_alloc((void**)&d_c, (size_t)tot * 4));
	const wtz_alnres_dev_t *dr = c->d_alnres;
	CHK(wtz_launch<K_pack_cigars>(0, c->n_items, [=] WTZ_LAMBDA (uint64_t t){ const wtz_alnres_dev_t &r = dr[t]; for(uint32_t k = 0; k < r.cigar_len; k++) d_c[d_off[t] + k] = r.cigar[k]; }));
	CHK(dev_sync());
	CHK(dev_d2h(dst, d_c, (size_t)tot * 4));
	dev_free(d_off); dev_free(d_c);
	return WTZ_OK;
}

/* the CIGAR text of the last wtz_pairs_align rendered into a device buffer of the context (grow-only, valid until the next call on this context) */
static int render_cigar_text(wtz_ctx_t *c, uint64_t n_bytes, char **d_text_out, bool wait = true){
	uint64_t tot = 0; for(uint32_t i = 0; i < c->n_items; i++) tot += c->h_alnres[i].text_len;
	if(tot != n_bytes) return wtz_fail(WTZ_E_ARG, "CIGAR text: expected room for %llu bytes, got %llu", (unsigned long long)tot, (unsigned long long)n_bytes);
	*d_text_out = NULL;
	if(tot == 0) return WTZ_OK;
#ifndef WTZ_EMUL
	{   /* the buffer may still be on its way out (the latest copy; the ones before it are in front of it on the same stream) */
		const uint64_t b = c->text_begun.load();
		if(b > c->text_known_done.load()){ HIPCHK(hipEventSynchronize(c->ev_text_done[(b - 1) & 1])); c->text_known_done.store(b); }
	}
#endif
	if(tot + 16 > c->cap_text){
		(void)dev_sync(); dev_free_persist(c->d_text); c->d_text = NULL; c->cap_text = 0;
		const size_t cap = (size_t)(tot + tot / 4 + 4096);
		CHK(dev_alloc_persist((void**)&c->d_text, cap)); c->cap_text = cap;
	}
	std::vector<uint64_t> off((size_t)c->n_items + 1);
	uint64_t o = 0; for(uint32_t i = 0; i < c->n_items; i++){ off[i] = o; o += c->h_alnres[i].text_len; } off[c->n_items] = o;
	uint64_t *d_off = NULL; char *d_t = c->d_text;
	CHK(dev_alloc((void**)&d_off, off.size() * 8)); CHK(dev_h2d(d_off, off.data(), off.size() * 8));
	const wtz_alnres_dev_t *dr = c->d_alnres;
	STAGE(c, "K_cigar_text");
	CHK(wtz_launch_coop<K_cigar_text>(0, c->n_items, [=] WTZ_LAMBDA (uint64_t t){ const wtz_alnres_dev_t &r = dr[t]; if(r.text_len) wtz_cigar_text_write_coop(r.cigar, r.cigar_len, d_t + d_off[t]); }));
	if(wait) CHK(dev_sync());
	dev_free(d_off);
	*d_text_out = d_t;
	return WTZ_OK;
}
extern "C" int wtz_fetch_cigar_text(wtz_ctx_t *c, char *dst, uint64_t n_bytes){
	if(!c || !c->have_items) return wtz_fail(WTZ_E_STATE, "wtz_fetch_cigar_text before wtz_pairs_align");
	CTX_ENTER(c);
	char *d_t = NULL;
	CHK(render_cigar_text(c, n_bytes, &d_t));
	if(n_bytes == 0) return WTZ_OK;
	if(!dst) return wtz_fail(WTZ_E_ARG, "null output");
	CHK(dev_d2h(dst, d_t, (size_t)n_bytes));
	return WTZ_OK;
}
/* the same in two halves: _begin renders the text and starts its copy on a stream of its own, _end waits for the copy.  Between the two the context is free for
 * the next calls (wtz_batch_begin ... wtz_pairs_align of the next range): at configs[2] the text is 3.4 GB per step = 72 ms at the rate of the link, and the
 * scratch pool is not involved - the text is rendered into a buffer of its own.  dst must stay valid (and untouched) until _end returns. */
extern "C" int wtz_fetch_cigar_text_begin(wtz_ctx_t *c, char *dst, uint64_t n_bytes){
	if(!c || !c->have_items) return wtz_fail(WTZ_E_STATE, "wtz_fetch_cigar_text_begin before wtz_pairs_align");
	CTX_ENTER(c);
#ifdef WTZ_EMUL
	char *d_t = NULL; CHK(render_cigar_text(c, n_bytes, &d_t));
	if(n_bytes && !dst) return wtz_fail(WTZ_E_ARG, "null output");
	if(n_bytes) memcpy(dst, d_t, (size_t)n_bytes);
	return WTZ_OK;
#else
	if(!c->stream_copy){
		if(hipStreamCreateWithFlags(&c->stream_copy, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_text_ready, hipEventDisableTiming) != hipSuccess
			|| hipEventCreateWithFlags(&c->ev_text_done[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_text_done[1], hipEventDisableTiming) != hipSuccess) return wtz_fail(WTZ_E_HIP, "hipStreamCreate / hipEventCreate failed");
	}
	char *d_t = NULL;
	CHK(render_cigar_text(c, n_bytes, &d_t, false));
	if(n_bytes == 0) return WTZ_OK;
	if(!dst) return wtz_fail(WTZ_E_ARG, "null output");
	HIPCHK(hipEventRecord(c->ev_text_ready, g_stream));
	HIPCHK(hipStreamWaitEvent(c->stream_copy, c->ev_text_ready, 0));
	HIPCHK(hipMemcpyAsync(dst, d_t, (size_t)n_bytes, hipMemcpyDeviceToHost, c->stream_copy));
	{ const uint64_t k = c->text_begun.load(); HIPCHK(hipEventRecord(c->ev_text_done[k & 1], c->stream_copy)); c->text_begun.store(k + 1); }
	return WTZ_OK;
#endif
}
extern "C" int wtz_fetch_cigar_text_end(wtz_ctx_t *c){
	if(!c) return wtz_fail(WTZ_E_ARG, "null argument");
#ifndef WTZ_EMUL
	/* no CTX_ENTER: this may be called while another thread runs the next range's calls on the context; it touches the event only */
	const uint64_t e = c->text_ended.load();
	if(e >= c->text_begun.load()) return WTZ_OK;                /* nothing in flight that has not been ended */
	if(e >= c->text_known_done.load()){ HIPCHK(hipEventSynchronize(c->ev_text_done[e & 1])); }      /* at worst the event has been re-recorded for copy e + 2 (whose render waited for copy e + 1): a longer wait, never a shorter one */
	c->text_ended.store(e + 1);
#endif
	return WTZ_OK;
}
extern "C" int wtz_cigar_text_device(wtz_ctx_t *c, uint64_t n_bytes, void **dev_ptr){
	if(!c || !c->have_items || !dev_ptr) return wtz_fail(WTZ_E_STATE, "wtz_cigar_text_device before wtz_pairs_align / null argument");
	CTX_ENTER(c);
	char *d_t = NULL;
	CHK(render_cigar_text(c, n_bytes, &d_t));
	*dev_ptr = d_t;
	return WTZ_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* f2: end extensions for a caller that holds its own overlaps (wtext)                                */
/* ------------------------------------------------------------------------------------------------ */
/* kswx_extend_align (kswx.h:469-481) = kswx_extend_align_shift_core (kswx.h:101-232) for n independent problems on views of the uploaded reads: the SAME job
 * dispatch as the ends of wtzmo's stitched alignments (run_extjobs: register DP on one or four wavefronts per job, LDS-ring and scalar forms for what is
 * outside their envelope).  out[i]: the kswx_t of the call + where its CIGAR words (traceback order reversed: first operation first) start in `cigar`. */
struct K_extcopy;
extern "C" int wtz_extend_batch(wtz_ctx_t *c, const wtz_dp_problem_t *pr, uint32_t n, wtz_dp_result_t *out, uint32_t *cigar, uint64_t cigar_cap){
	if(!c || !c->bits) return wtz_fail(WTZ_E_ARG, "reads not uploaded");
	if(n == 0) return WTZ_OK;
	if(!pr || !out || (!cigar && cigar_cap)) return wtz_fail(WTZ_E_ARG, "null argument");
	CTX_ENTER(c);
	CHK(pool_reset(c));
	std::vector<uint64_t> h_off(c->n_reads);
	CHK(dev_d2h(h_off.data(), c->rdoff, (size_t)c->n_reads * 8));
	std::vector<wtz_extjob_t> jobs(n);
	for(uint32_t i = 0; i < n; i++){
		const wtz_dp_problem_t &p = pr[i];
		if(p.q_read >= c->n_reads || p.t_read >= c->n_reads) return wtz_fail(WTZ_E_ARG, "problem %u: read id out of range", i);
		if((p.q_strand != 1 && p.q_strand != -1) || (p.t_strand != 1 && p.t_strand != -1)) return wtz_fail(WTZ_E_ARG, "problem %u: strand must be +1 or -1", i);
		wtz_readview vq, vt;
		vq.bits = c->bits; vq.off = h_off[p.q_read]; vq.len = c->h_rdlen[p.q_read]; vq.rev = p.q_rev ? 1u : 0u;
		vt.bits = c->bits; vt.off = h_off[p.t_read]; vt.len = c->h_rdlen[p.t_read]; vt.rev = p.t_rev ? 1u : 0u;
		const int64_t qlast = (int64_t)p.q_from + (int64_t)p.q_strand * (p.q_len > 0 ? p.q_len - 1 : 0), tlast = (int64_t)p.t_from + (int64_t)p.t_strand * (p.t_len > 0 ? p.t_len - 1 : 0);
		if(p.q_len < 0 || p.t_len < 0 || (p.q_len > 0 && (p.q_from < 0 || p.q_from >= (int64_t)vq.len || qlast < 0 || qlast >= (int64_t)vq.len))
				|| (p.t_len > 0 && (p.t_from < 0 || p.t_from >= (int64_t)vt.len || tlast < 0 || tlast >= (int64_t)vt.len)))
			return wtz_fail(WTZ_E_ARG, "problem %u: region outside its read", i);
		wtz_extjob_t j; memset(&j, 0, sizeof j);
		j.q = vq.sub(p.q_from, p.q_strand); j.t = vt.sub(p.t_from, p.t_strand); j.qlen = p.q_len; j.tlen = p.t_len; j.init_score = p.init_score; j.W = p.W; j.item = i; j.valid = 1;
		jobs[i] = j;
	}
	const wtz_env_t V = ctx_env(c);
	wtz_extjob_t *d_jobs = NULL; CHK(dev_alloc((void**)&d_jobs, (size_t)n * sizeof(wtz_extjob_t))); CHK(dev_h2d(d_jobs, jobs.data(), (size_t)n * sizeof(wtz_extjob_t)));
	wtz_timer tm; tm.start();
	CHK(run_extjobs(c, V, d_jobs, n));
	CHK(dev_sync());
	c->cnt.ms_stitch += tm.stop();
	CHK(tpool_check(c, "wtz_extend_batch"));
	CHK(dev_d2h(jobs.data(), d_jobs, (size_t)n * sizeof(wtz_extjob_t)));
	std::vector<uint64_t> off((size_t)n + 1); uint64_t tot = 0;
	for(uint32_t i = 0; i < n; i++){
		if(jobs[i].bad) return wtz_fail(WTZ_E_POOL, "wtz_extend_batch: problem %u ran out of scratch", i);
		const bool empty = jobs[i].qlen <= 0 || jobs[i].tlen <= 0;
		off[i] = tot; tot += empty ? 0 : jobs[i].cigar_len;
		c->cnt.cells_shift += jobs[i].cells;
	}
	off[n] = tot;
	if(tot > cigar_cap) return wtz_fail(WTZ_E_ARG, "wtz_extend_batch: CIGAR buffer too small (%llu words needed)", (unsigned long long)tot);
	if(tot){
		uint64_t *d_off = NULL; uint32_t *d_flat = NULL;
		CHK(dev_alloc((void**)&d_off, ((size_t)n + 1) * 8)); CHK(dev_h2d(d_off, off.data(), ((size_t)n + 1) * 8));
		CHK(dev_alloc((void**)&d_flat, (size_t)tot * 4));
		CHK(wtz_launch<K_extcopy>(0, n, [=] WTZ_LAMBDA (uint64_t t){ const uint64_t o = d_off[t], e = d_off[t + 1]; const uint32_t *src = d_jobs[t].cigar; for(uint64_t k = o; k < e; k++) d_flat[k] = src[k - o]; }));
		CHK(dev_sync());
		CHK(dev_d2h(cigar, d_flat, (size_t)tot * 4));
	}
	for(uint32_t i = 0; i < n; i++){
		wtz_dp_result_t o; memset(&o, 0, sizeof o);
		const wtz_extjob_t &j = jobs[i];
		const bool empty = j.qlen <= 0 || j.tlen <= 0;
		if(empty){ o.score = j.init_score < 0 ? 0 : j.init_score; }      /* kswx.h:113-118: an empty side returns the (clamped) start score and no operations */
		else { o.score = j.x.score; o.tb = j.x.tb; o.te = j.x.te; o.qb = j.x.qb; o.qe = j.x.qe; o.aln = j.x.aln; o.mat = j.x.mat; o.mis = j.x.mis; o.ins = j.x.ins; o.del = j.x.del; o.cigar_len = j.cigar_len; }
		o.cigar_off = off[i]; o.cells = j.cells; o.form_used = j.done ? j.done : 3;
		out[i] = o;
	}
	CHK(pool_check(c, "wtz_extend_batch"));
	return WTZ_OK;
}

#include "wtz_testdp.h"

extern "C" void *wtz_host_alloc(uint64_t n_bytes){
#ifdef WTZ_EMUL
	return malloc((size_t)(n_bytes ? n_bytes : 1));
#else
	void *p = NULL;
	if(hipHostMalloc(&p, (size_t)(n_bytes ? n_bytes : 1), hipHostMallocDefault) != hipSuccess){ (void)hipGetLastError(); return NULL; }
	return p;
#endif
}
extern "C" void wtz_host_free(void *p){
	if(!p) return;
#ifdef WTZ_EMUL
	free(p);
#else
	(void)hipHostFree(p);
#endif
}

extern "C" int wtz_pool_failure_kind(wtz_ctx_t *c){ return c ? c->last_pool_fail : 0; }

extern "C" int wtz_pool_info(wtz_ctx_t *c, wtz_pool_info_t *out){
	if(!c || !out) return wtz_fail(WTZ_E_ARG, "null argument");
	out->main_cap = c->main_bytes; out->main_used = c->main_used_call; out->transient_cap = c->pool_bytes - c->main_bytes; out->transient_peak = c->tpool_peak_call;
	return WTZ_OK;
}

extern "C" int wtz_get_counters(wtz_ctx_t *c, wtz_counters_t *out){
	if(!c || !out) return wtz_fail(WTZ_E_ARG, "null argument");
	*out = c->cnt;
#if !defined(WTZ_EMUL) && defined(WTZ_PROFILE)
	if(c->env_profile){        /* device phase profiler: Mticks per slot since the last report */
		unsigned long long h[64], z[64]; memset(z, 0, sizeof z);
		if(hipMemcpyFromSymbol(h, HIP_SYMBOL(wtz_prof), sizeof h) == hipSuccess){
			fprintf(stderr, "[phase-profile] Mticks:");
			for(int k = 0; k < 64; k++) fprintf(stderr, " %d:%.1f", k, (double)h[k] / 1e6);
			fprintf(stderr, "\n");
			(void)hipMemcpyToSymbol(HIP_SYMBOL(wtz_prof), z, sizeof z);
		}
	}
#endif
#if !defined(WTZ_EMUL) && defined(WTZ_PROFILE_CAND)
	if(c->env_profile){
		unsigned long long h[16], z[16]; memset(z, 0, sizeof z);
		if(hipMemcpyFromSymbol(h, HIP_SYMBOL(wtz_prof_cand), sizeof h) == hipSuccess){
			fprintf(stderr, "[cand-profile] Mticks / counts:");
			for(int k = 0; k < 16; k++) fprintf(stderr, " %d:%.1f", k, (double)h[k] / 1e6);
			fprintf(stderr, "\n");
			(void)hipMemcpyToSymbol(HIP_SYMBOL(wtz_prof_cand), z, sizeof z);
		}
	}
#endif
	return WTZ_OK;
}
extern "C" int wtz_reset_counters(wtz_ctx_t *c){ if(!c) return wtz_fail(WTZ_E_ARG, "null context"); memset(&c->cnt, 0, sizeof c->cnt); return WTZ_OK; }
