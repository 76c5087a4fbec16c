/*
 * wtz_ingest.h — f4: FASTA / FASTQ bases -> the reference's 2-bit BaseBank ON THE DEVICE.
 *
 *   wtz_pack_word      seq2basebank, dna.h:397-410 + base_bit_table dna.h:29-49 + bit2bits dna.h:78: 32 ASCII bases -> one word,
 *                      base i at bits ((~i)&31)*2 of word i>>5; A/a 0, C/c 1, G/g 2, T/t 3
 *   wtz_lrand48_state  every other byte becomes `lrand48() & 3` in FILE order (dna.h:405).  glibc's lrand48 is the 48-bit LCG
 *                      X(k+1) = 0x5DEECE66D * X(k) + 0xB; a process that never called srand48 starts from X(0) = 0 (glibc's state is
 *                      zero-initialised static data - NOT the 0x1234ABCD330E of the SVID text: its first four draws are 0, 2116118, 89401895,
 *                      379337186), and the k-th call returns X(k) >> 17.  The k-th non-ACGT base of the input therefore gets bits 17-18 of X(k): the pack kernel
 *                      only lists the positions of such bases, the host orders the (few) positions, and the fix-up kernel jumps the
 *                      LCG to each rank in O(log k) by squaring the affine map.
 *
 * HBM-bound by construction: 1 byte read + 2 bits written per base, no re-reads.
 */
#ifndef WTZ_INGEST_H
#define WTZ_INGEST_H

#include "wtz_common.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define WTZ_ING_NEXT(p) atomicAdd((p), 1ull)
#define WTZ_ING_OR(p, v) atomicOr((p), (v))
#else
#define WTZ_ING_NEXT(p) ((*(p))++)
#define WTZ_ING_OR(p, v) (*(p) |= (v))
#endif

/* X(k) of glibc's never-seeded drand48 family: the affine map x -> a*x + c (mod 2^48) applied k times to X(0) = 0 */
WTZ_HD uint64_t wtz_lrand48_state(uint64_t k){
	const uint64_t MASK = (1ull << 48) - 1;
	uint64_t A = 1, C = 0;                      /* accumulated map: identity */
	uint64_t a = 0x5DEECE66Dull, c = 0xBull;    /* current power of the step */
	while(k){
		if(k & 1){ C = (a * C + c) & MASK; A = (a * A) & MASK; }
		c = (a * c + c) & MASK; a = (a * a) & MASK;
		k >>= 1;
	}
	(void)A;                                    /* A * X(0) with X(0) = 0 */
	return C & MASK;
}

/* 2-bit code of one byte, 4 = not a base (dna.h:29-49: exactly A a C c G g T t are bases) */
WTZ_HD uint32_t wtz_base_code(uint32_t ch){
	const uint32_t lo = ch | 0x20u;
	const uint32_t t = (lo >> 1) & 3u;          /* a 0, c 1, g 3, t 2 */
	const bool ok = (lo == 0x61u) || (lo == 0x63u) || (lo == 0x67u) || (lo == 0x74u);
	return ok ? (t ^ (t >> 1)) : 4u;
}

/* four bytes at once: 0x80 in every byte of `v` that is one of ACGTacgt (exact zero-byte test, no borrow between bytes) */
WTZ_HD uint32_t wtz_base_mask4(uint32_t lo){
#define WTZ_EQ4(c) ({ const uint32_t x_ = lo ^ ((c) * 0x01010101u); ~(((x_ & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x_ | 0x7F7F7F7Fu); })
	return WTZ_EQ4(0x61u) | WTZ_EQ4(0x63u) | WTZ_EQ4(0x67u) | WTZ_EQ4(0x74u);
#undef WTZ_EQ4
}
/* the four 2-bit codes of four base bytes as one byte, first base in the top bits */
WTZ_HD uint32_t wtz_base_pack4(uint32_t lo){
	const uint32_t t = (lo >> 1) & 0x03030303u;                 /* a 0, c 1, g 3, t 2 per byte */
	const uint32_t code = t ^ ((t >> 1) & 0x01010101u);         /* a 0, c 1, g 2, t 3 */
	return (code * 0x40100401u) >> 24;                          /* c0 << 6 | c1 << 4 | c2 << 2 | c3: the partial products do not overlap */
}

/* HALF word `h` of the bank (16 bases: ascii[h*16 .. h*16+16), clipped at n) as it sits in the 64-bit word h >> 1: the even half holds the
 * word's first 16 bases = its UPPER 32 bits.  Positions of non-bases are appended to pos[] (any order; one reservation per call). */
WTZ_HD uint32_t wtz_pack_half(const uint8_t *ascii, uint64_t n, uint64_t h, unsigned long long *n_pos, uint64_t *pos, uint64_t pos_cap, uint64_t pos_base){
	const uint64_t b0 = h * 16;
	if(b0 + 16 <= n){
#if defined(__HIP_DEVICE_COMPILE__)
		const uint4 q = *(const uint4*)(const void*)(ascii + b0);             /* one 16-byte load: device chunks start 256-byte aligned, b0 is a multiple of 16 */
		const uint32_t v0 = q.x | 0x20202020u, v1 = q.y | 0x20202020u, v2 = q.z | 0x20202020u, v3 = q.w | 0x20202020u;
#else
		uint32_t p4[4]; memcpy(p4, ascii + b0, 16);
		const uint32_t v0 = p4[0] | 0x20202020u, v1 = p4[1] | 0x20202020u, v2 = p4[2] | 0x20202020u, v3 = p4[3] | 0x20202020u;
#endif
		if((wtz_base_mask4(v0) & wtz_base_mask4(v1) & wtz_base_mask4(v2) & wtz_base_mask4(v3)) == 0x80808080u)
			return (wtz_base_pack4(v0) << 24) | (wtz_base_pack4(v1) << 16) | (wtz_base_pack4(v2) << 8) | wtz_base_pack4(v3);
	}
	/* a non-base byte or the tail of the input: byte by byte */
	uint32_t half = 0, bad = 0;
	const uint64_t e = b0 + 16 <= n ? b0 + 16 : n;
	for(uint64_t i = b0; i < e; i++){
		const uint32_t code = wtz_base_code(ascii[i]);
		if(code < 4u) half |= code << ((15 - (int)(i - b0)) * 2); else bad++;
	}
	if(bad){
#if defined(__HIP_DEVICE_COMPILE__)
		unsigned long long at = atomicAdd(n_pos, (unsigned long long)bad);
#else
		unsigned long long at = *n_pos; *n_pos += bad;
#endif
		for(uint64_t i = b0; i < e; i++) if(wtz_base_code(ascii[i]) >= 4u){ if(at < pos_cap) pos[at] = pos_base + i; at++; }
	}
	return half;
}

/* the r-th listed position (ascending) is non-base number rank0 + r + 1 of the input: its two bits come from X(rank0 + r + 1) */
WTZ_HD void wtz_fix_random_base(uint64_t r, const uint64_t *pos_sorted, uint64_t rank0, uint64_t *bits){
	const uint64_t i = pos_sorted[r];
	const uint64_t v = (wtz_lrand48_state(rank0 + r + 1) >> 17) & 3ull;
	if(v) WTZ_ING_OR((unsigned long long*)&bits[i >> 5], (unsigned long long)(v << (((~i) & 31u) << 1)));
}

/* word `k` of the reverse-complement view of a read: view base j = 3 - read base (len - 1 - j)  (revbitseq_basebank, dna.h) */
WTZ_HD uint64_t wtz_revcomp_word(const uint64_t *bits, uint64_t off, uint32_t len, uint32_t k){
	uint64_t w = 0;
	const uint32_t j0 = k * 32u;
	for(uint32_t t = 0; t < 32u && j0 + t < len; t++){
		const uint32_t b = 3u - wtz_base_at(bits, off + (uint64_t)(len - 1u - (j0 + t)));
		w |= (uint64_t)b << ((31u - t) * 2u);
	}
	return w;
}

#endif
