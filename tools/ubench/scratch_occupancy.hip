// Resident wavefronts per SIMD of a 64-thread kernel with ~168 VGPRs, with and without a private (scratch) segment: does scratch cap the occupancy?
// hipcc --offload-arch=gfx950 -O3 -o scratch_occupancy scratch_occupancy.hip && ./scratch_occupancy
#include <hip/hip_runtime.h>
#include <stdio.h>
template<int SCR>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 8))) spin(unsigned *out, int iters, unsigned long long *t){
	volatile unsigned priv[SCR ? SCR : 1];
	if(SCR){ for(int i = 0; i < SCR; i++) priv[i] = threadIdx.x + i; }
	unsigned x = threadIdx.x;
	asm volatile("v_mov_b32 v160, %0" :: "v"(x) : "v160");      // forces ~161+ VGPRs
	for(int i = 0; i < iters; i++){ x = x * 1664525u + 1013904223u; }
	if(SCR) x += priv[x % SCR];      // the segment is touched before and after the loop only
	unsigned y; asm volatile("v_mov_b32 %0, v160" : "=v"(y) :: "v160");
	if(x == 0x12345678u) out[0] = x + y;
	if(threadIdx.x == 0 && blockIdx.x == 0) t[0] = 1;
}
template<int SCR> float run(int waves, int iters, unsigned *d, unsigned long long *t){
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipLaunchKernelGGL(spin<SCR>, dim3(waves), dim3(64), 8320, 0, d, iters, t); hipDeviceSynchronize();
	hipEventRecord(a); hipLaunchKernelGGL(spin<SCR>, dim3(waves), dim3(64), 8320, 0, d, iters, t); hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main(){
	unsigned *d; unsigned long long *t; hipMalloc(&d, 64); hipMalloc(&t, 64);
	const int iters = 200000;
	for(int per_simd = 1; per_simd <= 5; per_simd++){
		const int waves = 1024 * per_simd;
		printf("waves/SIMD offered %d: no scratch %.2f ms, 76 dwords of scratch %.2f ms, 130 dwords %.2f ms\n", per_simd, run<0>(waves, iters, d, t), run<76>(waves, iters, d, t), run<130>(waves, iters, d, t));
	}
	return 0;
}
