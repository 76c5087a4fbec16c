/*
 * Both end extensions of a stitched overlap on ONE wavefront (round 5).
 *
 * The stitch of hzm_aln.h:1304-1420 is sequential per pair: left extension (K-sw3, kswx.h:101-232) -> windows and gaps joined ->
 * right extension, whose init_score is the score so far.  Run as three launches per range (left jobs, K_stitch_mid, right jobs) every
 * extension launch ends in the tail of its longest job: a job's rows are sequential (~2 us each on a wavefront), the few jobs of
 * 3 000-5 000 rows take 7-9 ms whatever else the launch holds, and a range has only ~17 000-31 000 jobs of ~1 ms of wave time each for
 * 2 048 resident waves (tools/ubench/ksw3_bench.py: 4 000 / 17 000 / 40 000 jobs of a configs[2] step take 8.7 / 9.5 / 16.1 ms).
 * Here the wavefront that ran the left extension of an item goes on with the item's join (wtz_task_stitch_mid) and its right
 * extension: one launch and one tail per range instead of two, and twice the work per wave to hide it under.  An item whose
 * extension is outside the frame kernel's envelope leaves the launch where it stands; the launches behind it (general extension
 * kernel, K_stitch_mid for the items not marked, general kernel again) finish it exactly as before.
 */
#ifndef WTZ_STITCH_FUSED_H
#define WTZ_STITCH_FUSED_H

#include "wtz_tasks.h"
#include "wtz_sw_frame.h"
#include "wtz_sw_frame_mw.h"
#include "wtz_sw_frame16.h"

#ifdef __HIPCC__
/* the extension as a function of its own: its 250 registers are allocated for it alone, and what the kernel keeps across the call (the item, the side) is saved
 * once per job at the call, not spilled inside the row loop */
template<int TW>
__device__ __attribute__((noinline)) bool wtz_extjob_run_fr_call(wtz_extjob_t *job, const wtz_params_t *Pm, wtz_pool_t *pool, wtz_pool_t *tpool){
	__shared__ uint64_t stb[TW];      /* declared here, not passed in: the row loop reads it with LDS instructions, not through a generic pointer */
	return wtz_extjob_run_fr<TW, 0, 32>(job, Pm, pool, tpool, stb);
}
template<int TW>
__device__ __attribute__((noinline)) void wtz_stitch_mid_call(uint32_t t, const wtz_env_t *V, const wtz_alnitem_t *items, wtz_stitch_state_t *sts, const wtz_extjob_t *jobsL, wtz_extjob_t *jobsR, const wtz_gapres_t *gaps){
	wtz_task_stitch_mid(t, *V, items, sts, jobsL, jobsR, gaps, true);
}
template<int TW>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WTZ_OCC_EXTFR, 8))) wtz_kernel_stitch_ext_fr(const wtz_env_t V, const wtz_alnitem_t *items, wtz_stitch_state_t *sts,
		wtz_extjob_t *jobsL, wtz_extjob_t *jobsR, const wtz_gapres_t *gaps, const uint32_t *order, uint32_t n, uint32_t stride, uint32_t first){
	const uint32_t b = blockIdx.x;
	if(b >= n) return;
	const uint32_t t = order[(size_t)b * stride + first];      /* group `first` of `stride`: every stride-th item of the order, so that each group is ordered longest-first too */
	#pragma nounroll
	for(int side = 0; side < 2; side++){
		if(side){
			wtz_stitch_mid_call<TW>(t, &V, items, sts, jobsL, jobsR, gaps);
			__threadfence_block();      /* lane 0 wrote the right job; the whole wave reads it */
		}
		if(!wtz_extjob_run_fr_call<TW>(side ? &jobsR[t] : &jobsL[t], V.P, V.pool, V.pool + 1)) return;
		__threadfence_block();          /* lane 0 wrote the result and the operation list the join reads */
	}
}
/* the same launch in the packed 16-bit form (wtz_sw_frame16.h): a kernel of its own, because its register need (and with it the resident wavefronts per SIMD) is a
 * fraction of the 32-bit form's.  An item with an extension outside the 16-bit window leaves the launch where it stands and is listed in `open` (open[0] = count);
 * wtz_kernel_stitch_ext_fr over that list finishes it (wtz_task_stitch_mid and the jobs skip what is done). */
/* Both callees INLINED here, unlike in wtz_kernel_stitch_ext_fr: a register budget (amdgpu_waves_per_eu) binds the kernel's own code only - a function called from
 * it is compiled for itself and the kernel is allocated the maximum over its callees (the join came out at 179 registers in the library build, 167 in a build of
 * this kernel alone: two wavefronts per SIMD instead of three, measured as the same residency as the 32-bit form: SQ_WAVE_CYCLES / SQ_BUSY_CYCLES).  Inlined, the
 * three-wave budget holds for everything; what it costs is two or three spilled values per row of the widest class. */
#define WTZ_PK_LDS_BYTES(TW) ((TW) * 8 + 64)
template<int TW>
__device__ __attribute__((always_inline)) inline bool wtz_extjob_run_pk_call(wtz_extjob_t *job, const wtz_params_t *Pm, wtz_pool_t *pool, wtz_pool_t *tpool){
	/* the wave's dynamic LDS slice (the join between the extensions uses its first WTZ_WAVE_LDS_BYTES while no extension runs): WTZ_PK_LDS_BYTES per wave instead of
	 * a static array beside the slice - at 16 KB per wave a CU holds nine waves, and this kernel is built for twelve and more */
	return wtz_extjob_run_pk<TW>(job, Pm, pool, tpool, (uint64_t*)wtz_wave_scratch());
}
template<int TW>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WTZ_OCC_EXTPK, 8))) wtz_kernel_stitch_ext_pk(const wtz_env_t V, const wtz_alnitem_t *items, wtz_stitch_state_t *sts,
		wtz_extjob_t *jobsL, wtz_extjob_t *jobsR, const wtz_gapres_t *gaps, const uint32_t *order, uint32_t n, uint32_t stride, uint32_t first, uint32_t *open){
	const uint32_t b = blockIdx.x;
	if(b >= n) return;
	const uint32_t t = order[(size_t)b * stride + first];
	#pragma nounroll
	for(int side = 0; side < 2; side++){
		if(side){
			wtz_task_stitch_mid(t, V, items, sts, jobsL, jobsR, gaps, true);
			__threadfence_block();
		}
		if(!wtz_extjob_run_pk_call<TW>(side ? &jobsR[t] : &jobsL[t], V.P, V.pool, V.pool + 1)){
			if((threadIdx.x & 63u) == 0){ const uint32_t k = atomicAdd(&open[0], 1u); open[1 + k] = t; }
			return;
		}
		__threadfence_block();
	}
}
/* the same for the items with the longest extensions, on FOUR wavefronts per item (wtz_sw_frame_mw.h): these few items are the critical path of the whole launch.
 * Wave 0 runs the join between the two extensions; the other waves wait at the barrier. */
template<int TW>
__global__ void __launch_bounds__(256) wtz_kernel_stitch_ext_frmw(const wtz_env_t V, const wtz_alnitem_t *items, wtz_stitch_state_t *sts,
		wtz_extjob_t *jobsL, wtz_extjob_t *jobsR, const wtz_gapres_t *gaps, const uint32_t *order, uint32_t n){
	__shared__ uint64_t stb[TW]; __shared__ wtz_frmw_shared_t shm;
	const uint32_t b = blockIdx.x;
	if(b >= n) return;
	const uint32_t t = order[b];
	#pragma nounroll
	for(int side = 0; side < 2; side++){
		if(side){
			if(threadIdx.x < 64u) wtz_stitch_mid_call<TW>(t, &V, items, sts, jobsL, jobsR, gaps);
			__threadfence_block(); __syncthreads();      /* lane 0 wrote the right job; the whole workgroup reads it */
		}
		if(!wtz_extjob_run_frmw<TW>(side ? &jobsR[t] : &jobsL[t], V.P, V.pool, V.pool + 1, stb, &shm)) return;      /* the same answer in every thread */
		__threadfence_block(); __syncthreads();
	}
}
#endif

#endif
