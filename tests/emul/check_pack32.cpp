#include <cstdio>
#include <cstdlib>
#include <cstring>
#define WTZ_EMUL 1
#include "wtz_sw.h"
int main(){
	srand48(5);
	const int NB = 5000; const int NW = NB/32 + 2;
	uint64_t *bits = (uint64_t*)calloc(NW, 8);
	for(int i = 0; i < NW; i++) bits[i] = ((uint64_t)lrand48() << 33) ^ ((uint64_t)lrand48() << 11) ^ (uint64_t)lrand48();
	long bad = 0, n = 0;
	for(int it = 0; it < 200000; it++){
		wtz_seq_packed s; s.bits = bits; s.strand = (lrand48() & 1) ? 1 : -1; s.comp = lrand48() & 1;
		int len = 1 + lrand48() % 300;
		if(s.strand > 0) s.start = lrand48() % (NB - len); else s.start = len - 1 + lrand48() % (NB - len);
		int b0 = (lrand48() % ((len + 31) / 32 + 1)) * 32;
		uint64_t ref = 0; for(int k = 0; k < 32 && b0 + k < len; k++) ref |= ((uint64_t)s.at(b0 + k)) << (2 * k);
		uint64_t got = wtz_pack32(s, b0, len);
		n++; if(ref != got){ if(bad < 5) printf("MISMATCH strand %d comp %u start %ld len %d b0 %d ref %016lx got %016lx\n", s.strand, s.comp, (long)s.start, len, b0, ref, got); bad++; }
	}
	printf("%ld tests, %ld bad\n", n, bad);
	return bad != 0;
}
