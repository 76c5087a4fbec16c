/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see ora_util.h).
 * liboracle.so: function-level entry points of the CPU restatement for ctypes-driven tests.
 */
#include "ora_overlap.h"

typedef ora_aln_t shim_aln_t;

int ora_extend_fixed_c(int qlen, uint8_t *q, int tlen, uint8_t *t, int strand, int init_score, int W, int M, int X, int I, int D, int E, int T, shim_aln_t *out, uint32_t *cigar_out){
	ora_swmem_t mem; memset(&mem, 0, sizeof mem); vec_u32 cg = {0};
	*out = ora_extend_fixed(qlen, q, tlen, t, strand, init_score, W, M, X, I, D, E, T, &mem, &cg);
	int n = (int)cg.n; memcpy(cigar_out, cg.a, 4 * cg.n);
	vec_i32_free(&mem.rh); vec_i32_free(&mem.re); vec_i32_free(&mem.zb); vec_u8_free(&mem.z); vec_u32_free(&cg);
	return n;
}
int ora_extend_shift_c(int qlen, uint8_t *q, int tlen, uint8_t *t, int strand, int init_score, int W, int M, int X, int I, int D, int E, int T, shim_aln_t *out, uint32_t *cigar_out){
	ora_swmem_t mem; memset(&mem, 0, sizeof mem); vec_u32 cg = {0};
	*out = ora_extend_shift(qlen, q, tlen, t, strand, init_score, W, M, X, I, D, E, T, &mem, &cg);
	int n = (int)cg.n; memcpy(cigar_out, cg.a, 4 * cg.n);
	vec_i32_free(&mem.rh); vec_i32_free(&mem.re); vec_i32_free(&mem.zb); vec_u8_free(&mem.z); vec_u32_free(&cg);
	return n;
}
int ora_global_c(int qlen, uint8_t *q, int tlen, uint8_t *t, int M, int X, int o_del, int e_del, int o_ins, int e_ins, int w, int *score, uint32_t *cigar_out){
	int8_t mat[16]; for(int i = 0; i < 16; i++) mat[i] = ((i % 4) == (i / 4)) ? (int8_t)M : (int8_t)X;
	ora_swmem_t mem; memset(&mem, 0, sizeof mem); vec_u32 cg = {0};
	*score = ora_global_banded(qlen, q, tlen, t, mat, o_del, e_del, o_ins, e_ins, w, &mem, &cg);
	int n = (int)cg.n; memcpy(cigar_out, cg.a, 4 * cg.n);
	vec_i32_free(&mem.rh); vec_i32_free(&mem.re); vec_i32_free(&mem.zb); vec_u8_free(&mem.z); vec_u32_free(&cg);
	return n;
}
void ora_sort_u64_lo32_desc(uint64_t *v, size_t n){ ora_sort_cand_desc(v, n, NULL); }
#define ORA_U32_GT(a, b) ((a) > (b))
ORA_DEFINE_SORT(ora_sort_u32_asc_impl, uint32_t, ORA_U32_GT)
void ora_sort_u32_asc(uint32_t *v, size_t n){ ora_sort_u32_asc_impl(v, n, NULL); }
int ora_median_c(int32_t *v, int32_t n){ return ora_median(v, n); }
