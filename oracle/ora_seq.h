/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see ora_util.h).
 *
 * ora_seq.h — read ingest: FASTA/FASTQ parsing, the 2-bit sequence store and the read table.
 * Restates:
 *   - record parsing            reference file_reader.c:296-345 (FASTA), 347-396 (FASTQ),
 *                               399-416 (type guess), 66-71 (`gzip -dc` popen for *.gz)
 *   - 2-bit packing             reference dna.h:78 (`bits2bit`), 263 (`bit2bits`), 397-410
 *                               (`seq2basebank`: non-ACGT -> lrand48()&3 in file order)
 *   - reverse complement k-mer  reference dna.h:85-98 (`dna_rev_seq`)
 *   - read table + length sort  reference wtzmo.c:87-90, 207-215, 1708 (unstable sort: the
 *                               read id is the rank under ora_util.h's exact sort)
 */
#ifndef ORA_SEQ_H
#define ORA_SEQ_H

#include "ora_util.h"

typedef struct {
	uint64_t off;     /* offset (bases) of the read inside the 2-bit store: rdoff:40 */
	uint32_t len;     /* rdlen:24 */
	char    *name;
} ora_read_t;
ORA_VEC(vec_read, ora_read_t)

typedef struct {
	uint64_t *bits;   /* 32 bases per word, base i at bit ((~i)&31)*2 of word i>>5 */
	uint64_t  nbase, capw;
	vec_read  reads;  /* after ora_sort_reads(): index == read id */
	uint32_t  n_rd;
} ora_store_t;

static inline uint32_t ora_base_at(const uint64_t *bits, uint64_t i){
	return (uint32_t)((bits[i >> 5] >> (((~i) & 31u) << 1)) & 3u);
}

static inline void ora_store_put(ora_store_t *st, uint32_t b){
	uint64_t i = st->nbase;
	if((i >> 5) >= st->capw){
		uint64_t c = st->capw ? st->capw * 2 : 1024;
		st->bits = (uint64_t*)ora_xrealloc(st->bits, c * 8);
		memset(st->bits + st->capw, 0, (c - st->capw) * 8);
		st->capw = c;
	}
	st->bits[i >> 5] |= ((uint64_t)(b & 3u)) << (((~i) & 31u) << 1);
	st->nbase = i + 1;
}

static inline int ora_code_of(int ch){
	switch(ch){
		case 'A': case 'a': return 0;
		case 'C': case 'c': return 1;
		case 'G': case 'g': return 2;
		case 'T': case 't': return 3;
		default: return 4;
	}
}

static inline void ora_store_add_read(ora_store_t *st, const char *name, size_t nlen, const char *seq, size_t slen){
	ora_read_t r;
	r.off = st->nbase; r.len = (uint32_t)slen;
	r.name = (char*)ora_xrealloc(NULL, nlen + 1);
	memcpy(r.name, name, nlen); r.name[nlen] = 0;
	vec_read_push(&st->reads, r);
	for(size_t i = 0; i < slen; i++){
		int c = ora_code_of((unsigned char)seq[i]);
		if(c == 4) c = (int)(lrand48() & 3);   /* dna.h:405, glibc default seed, file order */
		ora_store_put(st, (uint32_t)c);
	}
	st->n_rd++;
}

/* unpack [off, off+len) as one byte per base; rev!=0 gives the reverse complement (dna.h:463-475) */
static inline void ora_unpack(const ora_store_t *st, uint64_t off, uint32_t len, int rev, uint8_t *dst){
	if(!rev){ for(uint32_t i = 0; i < len; i++) dst[i] = (uint8_t)ora_base_at(st->bits, off + i); }
	else    { for(uint32_t i = 0; i < len; i++) dst[i] = (uint8_t)((~ora_base_at(st->bits, off + len - 1 - i)) & 3u); }
}

/* reverse complement of a k-mer held in the low 2k bits (dna.h:85-98) */
static inline uint64_t ora_revcomp_kmer(uint64_t x, unsigned k){
	x = ~x;
	x = ((x & 0x3333333333333333ULL) << 2) | ((x & 0xCCCCCCCCCCCCCCCCULL) >> 2);
	x = ((x & 0x0F0F0F0F0F0F0F0FULL) << 4) | ((x & 0xF0F0F0F0F0F0F0F0ULL) >> 4);
	x = __builtin_bswap64(x);
	return x >> (64 - (k << 1));
}

/* ---- line reader over a list of files (plain or .gz through `gzip -dc`) ---- */
typedef struct {
	char **paths; int npath, cur;
	FILE *fp; int is_proc;
	char *line; size_t cap; long n;     /* current line (no '\n'), n = length or -1 */
	int pushed;                          /* one-line roll back */
	int kind;                            /* 0 unknown, 1 fasta, 2 fastq */
} ora_reader_t;

static inline int ora_reader_open_cur(ora_reader_t *r){
	while(r->cur < r->npath){
		const char *p = r->paths[r->cur];
		size_t L = strlen(p);
		if(strcmp(p, "-") == 0){ r->fp = stdin; r->is_proc = 0; return 1; }
		if(L > 3 && strcmp(p + L - 3, ".gz") == 0){
			char *cmd = (char*)ora_xrealloc(NULL, L + 32);
			sprintf(cmd, "gzip -dc %s", p);
			r->fp = popen(cmd, "r"); r->is_proc = 1; free(cmd);
		} else { r->fp = fopen(p, "r"); r->is_proc = 0; }
		if(r->fp) return 1;
		return 0;
	}
	return 0;
}

static inline ora_reader_t *ora_reader_open(char **paths, int npath){
	ora_reader_t *r = (ora_reader_t*)calloc(1, sizeof(*r));
	r->paths = paths; r->npath = npath; r->cur = 0;
	if(!ora_reader_open_cur(r)){ free(r); return NULL; }
	return r;
}

static inline void ora_reader_close(ora_reader_t *r){
	if(r->fp && r->fp != stdin){ if(r->is_proc) pclose(r->fp); else fclose(r->fp); }
	free(r->line); free(r);
}

/* next line of the concatenated byte stream; returns length or -1 at end */
static inline long ora_reader_line(ora_reader_t *r){
	if(r->pushed){ r->pushed = 0; return r->n; }
	size_t n = 0; int got = 0;
	for(;;){
		if(r->fp == NULL) break;
		int c = fgetc(r->fp);
		if(c == EOF){
			if(r->fp != stdin){ if(r->is_proc) pclose(r->fp); else fclose(r->fp); }
			r->fp = NULL; r->cur++;
			if(r->cur < r->npath){ if(!ora_reader_open_cur(r)) r->fp = NULL; continue; }
			break;
		}
		got = 1;
		if(c == '\n') { r->n = (long)n; if(r->cap < n + 1){ r->cap = n + 1; r->line = (char*)ora_xrealloc(r->line, r->cap); } r->line[n] = 0; return r->n; }
		if(n + 2 > r->cap){ r->cap = r->cap ? r->cap * 2 : 256; r->line = (char*)ora_xrealloc(r->line, r->cap); }
		r->line[n++] = (char)c;
	}
	if(!got){ r->n = -1; return -1; }
	if(r->cap < n + 1){ r->cap = n + 1; r->line = (char*)ora_xrealloc(r->line, r->cap); }
	r->line[n] = 0; r->n = (long)n; return r->n;
}

/* one sequence record; name = header up to the first blank (file_reader.c:311-323) */
static inline int ora_reader_seq(ora_reader_t *r, vec_u8 *name, vec_u8 *seq){
	long n;
	if(r->kind == 0){
		while((n = ora_reader_line(r)) != -1){
			if(n == 0) continue;
			if(r->line[0] == '#') continue;
			r->kind = r->line[0] == '>' ? 1 : (r->line[0] == '@' ? 2 : 3);
			r->pushed = 1;
			break;
		}
		if(r->kind == 0) r->kind = 3;
	}
	name->n = 0; seq->n = 0;
	if(r->kind == 1){
		int flag = 0;
		while((n = ora_reader_line(r)) != -1){
			if(n && r->line[0] == '>'){
				if(flag){ r->pushed = 1; break; }
				flag = 1;
				long i;
				for(i = 1; i < n; i++){ char c = r->line[i]; if(c == ' ' || c == '\t' || c == '\r' || c == '\n') break; }
				vec_u8_append(name, (uint8_t*)r->line + 1, (size_t)(i - 1));
			} else if(flag){
				vec_u8_append(seq, (uint8_t*)r->line, (size_t)n);
				flag = 2;
			}
		}
		return flag != 0;
	} else if(r->kind == 2){
		int flag = 0;
		while(flag != 4 && (n = ora_reader_line(r)) >= 0){
			switch(flag){
				case 0:
					if(r->line[0] != '@') break;
					flag = 1;
					{ long i; for(i = 1; i < n; i++){ char c = r->line[i]; if(c == ' ' || c == '\t' || c == '\n') break; }
					  vec_u8_append(name, (uint8_t*)r->line + 1, (size_t)(i - 1)); }
					break;
				case 1: flag = 2; vec_u8_append(seq, (uint8_t*)r->line, (size_t)n); break;
				case 2: if(r->line[0] != '+') break; flag = 3; break;
				case 3: flag = 4; break;
			}
		}
		return flag == 4;
	}
	return 0;
}

/* wtzmo.c:1708 — read id := rank by length DESC under the exact unstable sort */
#define ORA_READ_GT(a, b) ((b).len > (a).len)
ORA_DEFINE_SORT(ora_sort_reads_by_len, ora_read_t, ORA_READ_GT)

#endif
