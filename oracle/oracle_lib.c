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

/* ---- f1: align_hzmaux (hzm_aln.h:1684-1775), whole-read form as wtgbo calls it ---- */
#include "ora_hzmaux.h"
int ora_align_hzmaux_c(const uint8_t *tseq, int tlen, const uint8_t *rdseq, int rdlen, const int32_t *prm, float min_sm, int refine, shim_aln_t *out, uint32_t *cigar_out, int cigar_cap){
	static ora_hzmaux_t A;     /* scratch kept across calls */
	ora_auxparams_t P;
	P.zsize = (uint32_t)prm[0]; P.hz = (uint32_t)prm[1]; P.zwin = (uint32_t)prm[2]; P.zstep = (uint32_t)prm[3]; P.zovl = (uint32_t)prm[4]; P.zmax = (uint32_t)prm[5]; P.zvar = (uint32_t)prm[6];
	P.w = prm[7]; P.W = prm[8]; P.ew = prm[9]; P.rw = prm[10]; P.M = prm[11]; P.X = prm[12]; P.I = prm[13]; P.D = prm[14]; P.E = prm[15]; P.T = prm[16];
	ora_hzmaux_index(&A, &P, tseq, (uint32_t)tlen);
	if(!ora_align_hzmaux(&A, &P, rdseq, rdlen, refine, min_sm)) return -1;
	*out = A.hit;
	if((int)A.cigars.n > cigar_cap) return -2;
	memcpy(cigar_out, A.cigars.a, 4 * A.cigars.n);
	return (int)A.cigars.n;
}
