/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * ref_shim.c — exports a few of the reference's own `static inline` routines from a shared object so that tests can
 * call the REAL reference function-by-function.  This file contains no reference code: it only
 * #includes the reference headers where they lie (compiled with `-iquote /root/reference` by oracle/Makefile into
 * oracle/_ref/libref_shim.so, which is git-ignored and travels to the GPU box prebuilt).
 */
#include "hzm_aln.h"

typedef struct { int score, tb, te, qb, qe, aln, mat, mis, ins, del; } shim_aln_t;

static shim_aln_t to_shim(kswx_t x){ shim_aln_t r = {x.score, x.tb, x.te, x.qb, x.qe, x.aln, x.mat, x.mis, x.ins, x.del}; return r; }

/* cigar_out must have room for qlen+tlen+2 words; returns number of words */
int ref_extend_fixed(int qlen, uint8_t *q, int tlen, uint8_t *t, int strand, int init_score, int W, int M, int X, int I, int D, int E, int T, shim_aln_t *out, uint32_t *cigar_out){
	u8list *mem = init_u8list(1024); u32list *cg = init_u32list(64);
	kswx_t x = kswx_extend_align_core(qlen, q, tlen, t, strand, init_score, W, M, X, I, D, E, T, mem, cg);
	*out = to_shim(x);
	int n = (int)cg->size; memcpy(cigar_out, cg->buffer, sizeof(uint32_t) * cg->size);
	free_u8list(mem); free_u32list(cg);
	return n;
}

int ref_extend_shift(int qlen, uint8_t *q, int tlen, uint8_t *t, int strand, int init_score, int W, int M, int X, int I, int D, int E, int T, shim_aln_t *out, uint32_t *cigar_out){
	u8list *mem = init_u8list(1024); u32list *cg = init_u32list(64);
	kswx_t x = kswx_extend_align_shift_core(qlen, q, tlen, t, strand, init_score, W, M, X, I, D, E, T, mem, cg);
	*out = to_shim(x);
	int n = (int)cg->size; memcpy(cigar_out, cg->buffer, sizeof(uint32_t) * cg->size);
	free_u8list(mem); free_u32list(cg);
	return n;
}

int ref_global(int qlen, uint8_t *q, int tlen, uint8_t *t, int M, int X, int o_del, int e_del, int o_ins, int e_ins, int w, int *score, uint32_t *cigar_out){
	int8_t mat[16]; int i, n = 0; uint32_t *cg = NULL;
	for(i = 0; i < 16; i++) mat[i] = ((i % 4) == (i / 4)) ? M : X;
	*score = ksw_global2(qlen, q, tlen, t, 4, mat, o_del, e_del, o_ins, e_ins, w, &n, &cg);
	if(n) memcpy(cigar_out, cg, sizeof(uint32_t) * n);
	free(cg);
	return n;
}

/* the unstable sort on u64 keyed by the low 32 bits descending (wtzmo.c:821) and ascending on the whole value */
void ref_sort_u64_lo32_desc(uint64_t *v, size_t n){ sort_array(v, n, uint64_t, (b & 0xFFFFFFFFU) > (a & 0xFFFFFFFFU)); }
void ref_sort_u32_asc(uint32_t *v, size_t n){ sort_array(v, n, uint32_t, a > b); }
int ref_median(int32_t *v, int32_t n){ return calculate_median_value(v, n); }

/* ---- f1: the reference's own align_hzmaux (hzm_aln.h:1684-1775) on one (target, read) pair, whole-read form as wtgbo calls it (wtgbo.c:37-56).
 * prm = zsize hz zwin zstep zovl zmax zvar w W ew rw M X I D E T.  Returns the number of CIGAR words, -1 = no hit. */
int ref_align_hzmaux(const uint8_t *tseq, int tlen, uint8_t *rdseq, int rdlen, const int32_t *prm, float min_sm, int refine, shim_aln_t *out, uint32_t *cigar_out, int cigar_cap){
	static HZMAux *aux = NULL;
	int i;
	if(aux == NULL) aux = init_hzmaux();
	aux->zsize = prm[0]; aux->hz = prm[1]; aux->zwin = prm[2]; aux->zstep = prm[3]; aux->zovl = prm[4]; aux->zmax = prm[5]; aux->zvar = prm[6];
	aux->w = prm[7]; aux->W = prm[8]; aux->ew = prm[9]; aux->rw = prm[10]; aux->M = prm[11]; aux->X = prm[12]; aux->I = prm[13]; aux->D = prm[14]; aux->E = prm[15]; aux->T = prm[16];
	aux->has_alignment = 0;
	reset_hzmaux(aux);
	for(i = 0; i < tlen; i++) add_tseq_hzmaux(aux, tseq[i]);
	ready_hzmaux(aux);
	if(!align_hzmaux(aux, 0, rdseq, NULL, rdlen, 0, 0, refine, min_sm)) return -1;
	*out = to_shim(aux->hit);
	if((int)aux->cigars->size > cigar_cap) return -2;
	memcpy(cigar_out, aux->cigars->buffer, sizeof(uint32_t) * aux->cigars->size);
	return (int)aux->cigars->size;
}

/* ---- f4: the reference's own loader of one sequence into a BaseBank (seq2basebank, dna.h:397-410) from the lrand48 state of a fresh glibc process (all
 * zero: seed48 of three zeros restores exactly that, the multiplier and the addend being reset to their defaults), after `skip` earlier lrand48 calls.  bits_out must hold (len + 31) / 32 + 1 words. */
void ref_seq2basebank(char *seq, uint64_t len, uint64_t skip, uint64_t *bits_out){
	unsigned short s0[3] = {0, 0, 0};
	uint64_t i;
	BaseBank *bnk = init_basebank();
	seed48(s0);
	for(i = 0; i < skip; i++) (void)lrand48();
	seq2basebank(bnk, seq, len);
	memcpy(bits_out, bnk->bits, ((len + 31) / 32) * 8);
	free_basebank(bnk);
}
