// Integer VALU issue rate of gfx950 ("calibrate the int32 roof with a v_add_u32 / v_max_i32 micro-benchmark").
// Every thread runs ITERS x 16 independent (or chained) ops of one kind on 16 accumulators; W waves per SIMD resident; the rate is
// reported as cycles per wave64 instruction per SIMD and as Tint32op/s for the whole chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITERS 4096
template<int KIND, int CHAIN> __global__ void __launch_bounds__(64) k(unsigned *out, unsigned seed){
	unsigned a[16];
	#pragma unroll
	for(int i = 0; i < 16; i++) a[i] = seed + threadIdx.x * 17u + i;
	unsigned b = seed ^ 0x9E3779B9u, c3 = seed | 1u;
	for(int it = 0; it < ITERS; it++){
		#pragma unroll
		for(int i = 0; i < 16; i++){
			unsigned &x = a[CHAIN ? 0 : i];
			if(KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
			if(KIND == 1) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x) : "v"(b));
			if(KIND == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(x) : "v"(b));
			if(KIND == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b));
			if(KIND == 4) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(x) : "v"(b), "v"(c3));
			if(KIND == 5) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(b), "v"(c3));
			if(KIND == 6) asm volatile("v_lshl_or_b32 %0, %0, 7, %1" : "+v"(x) : "v"(b));
			if(KIND == 7) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(b));
		}
	}
	unsigned s = 0;
	#pragma unroll
	for(int i = 0; i < 16; i++) s ^= a[i];
	if(s == 0x12345678u) out[0] = s;
}
template<int KIND, int CHAIN> static void run(const char *name, int cus, double ghz){
	unsigned *out; hipMalloc(&out, 4);
	for(int wps = 1; wps <= 8; wps *= 2){
		const int nblk = cus * 4 * wps;
		hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
		hipLaunchKernelGGL((k<KIND, CHAIN>), dim3(nblk), dim3(64), 0, 0, out, 1u);
		hipDeviceSynchronize();
		hipEventRecord(a);
		hipLaunchKernelGGL((k<KIND, CHAIN>), dim3(nblk), dim3(64), 0, 0, out, 1u);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		const double instr_per_simd = (double)ITERS * 16 * wps;
		const double cyc = ms * 1e-3 * ghz * 1e9 / instr_per_simd;
		printf("%-16s %s  %d wave(s)/SIMD: %.3f ms, %.2f cycles per wave64 instruction per SIMD, %.1f Tlane-op/s chip\n", name, CHAIN ? "dependent  " : "independent", wps, ms, cyc, (double)nblk * 64 * ITERS * 16 / (ms * 1e-3) / 1e12);
	}
	hipFree(out);
}
int main(){
	hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
	const double ghz = pr.clockRate / 1e6;
	printf("CUs %d, clock %.2f GHz\n", pr.multiProcessorCount, ghz);
	run<0, 0>("v_add_u32", pr.multiProcessorCount, ghz); run<0, 1>("v_add_u32", pr.multiProcessorCount, ghz);
	run<1, 0>("v_max_i32", pr.multiProcessorCount, ghz); run<1, 1>("v_max_i32", pr.multiProcessorCount, ghz);
	run<2, 0>("v_alignbit_b32", pr.multiProcessorCount, ghz);
	run<3, 0>("v_cndmask_b32", pr.multiProcessorCount, ghz);
	run<4, 0>("v_mad_i32_i24", pr.multiProcessorCount, ghz);
	run<6, 0>("v_lshl_or_b32", pr.multiProcessorCount, ghz);
	run<7, 0>("v_sub_u32", pr.multiProcessorCount, ghz);
	run<5, 0>("v_fma_f32", pr.multiProcessorCount, ghz); run<5, 1>("v_fma_f32", pr.multiProcessorCount, ghz);
	return 0;
}
