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
			/* round 5: the ops a packed 16-bit / select-free K-sw3 cell would be made of */
			if(KIND == 8) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(x) : "v"(b) : "s20", "s21");
			if(KIND == 9) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c3));
			if(KIND == 10) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x) : "v"(b));
			if(KIND == 11) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(x) : "v"(b));
			if(KIND == 12) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(x) : "v"(b));
			if(KIND == 13) asm volatile("v_bfe_u32 %0, %0, 3, 1" : "+v"(x));
			if(KIND == 14) asm volatile("v_lshl_add_u32 %0, %0, 11, %1" : "+v"(x) : "v"(b));
			if(KIND == 15) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c3));
			if(KIND == 16) asm volatile("v_cmp_lt_i32 s[22:23], %0, %1" : : "v"(x), "v"(b) : "s22", "s23");
			if(KIND == 17) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b));
			if(KIND == 18) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b));
			if(KIND == 19) asm volatile("v_pk_lshrrev_b16 %0, 15, %0" : "+v"(x));
			if(KIND == 20) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x) : "v"(b), "v"(c3));
			if(KIND == 21) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c3));
			if(KIND == 22) asm volatile("v_pk_mad_i16 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c3));
			if(KIND == 23) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c3));
			if(KIND == 24) asm volatile("v_lshrrev_b32 %0, 15, %0" : "+v"(x));
			if(KIND == 25) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b));
			if(KIND == 26) asm volatile("v_pk_ashrrev_i16 %0, 15, %0" : "+v"(x));
			if(KIND == 27) asm volatile("v_max_i32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b));
			if(KIND == 28) asm volatile("v_sub_u32 %0, %0, %1\n\tv_max_i32 %0, %0, %2" : "+v"(x) : "v"(b), "v"(c3));     /* a full-rate + a half-rate op: do they pair? (2 instructions) */
			if(KIND == 29) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b));
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
	run<8, 0>("v_cndmask(sgpr)", pr.multiProcessorCount, ghz); run<9, 0>("v_max3_i32", pr.multiProcessorCount, ghz);
	run<10, 0>("v_pk_max_i16", pr.multiProcessorCount, ghz); run<10, 1>("v_pk_max_i16", pr.multiProcessorCount, ghz);
	run<11, 0>("v_pk_add_i16", pr.multiProcessorCount, ghz); run<12, 0>("v_pk_sub_i16", pr.multiProcessorCount, ghz);
	run<13, 0>("v_bfe_u32", pr.multiProcessorCount, ghz); run<14, 0>("v_lshl_add_u32", pr.multiProcessorCount, ghz); run<15, 0>("v_and_or_b32", pr.multiProcessorCount, ghz);
	run<16, 0>("v_cmp_lt_i32", pr.multiProcessorCount, ghz); run<17, 0>("v_mov_b32", pr.multiProcessorCount, ghz); run<18, 0>("v_mov_b32_dpp", pr.multiProcessorCount, ghz);
	run<19, 0>("v_pk_lshrrev_b16", pr.multiProcessorCount, ghz); run<20, 0>("v_bfi_b32", pr.multiProcessorCount, ghz); run<21, 0>("v_perm_b32", pr.multiProcessorCount, ghz);
	run<22, 0>("v_pk_mad_i16", pr.multiProcessorCount, ghz); run<23, 0>("v_add3_u32", pr.multiProcessorCount, ghz); run<24, 0>("v_lshrrev_b32", pr.multiProcessorCount, ghz);
	run<25, 0>("v_and_b32", pr.multiProcessorCount, ghz); run<26, 0>("v_pk_ashrrev_i16", pr.multiProcessorCount, ghz); run<27, 0>("v_max_i32_dpp", pr.multiProcessorCount, ghz);
	run<28, 0>("sub+max (2 ops)", pr.multiProcessorCount, ghz); run<29, 0>("v_xor_b32", pr.multiProcessorCount, ghz);
	return 0;
}
