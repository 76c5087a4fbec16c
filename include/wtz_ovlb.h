/*
 * wtz_ovlb.h — binary overlap records: the hand-off between `wtzmo` / `wtgbo` and the programs that load their `.ovl` output
 * (SURVEY.md 8f3).  Header-only, plain C, no dependencies: a consumer includes it and replaces its text loader's inner loop.
 *
 * What it replaces.  The reference hands overlaps over as 17-column text (writer: print_hits_wtzmo, wtzmo.c:1235-1244) and every
 * consumer parses that text back into integers:
 *     wtlay / wtgbo   parse_overlap_item_strgraph   wtlay.h:238-268   (columns 0-12 of >= 16) inside the loader loop wtlay.h:443-470
 *     wtclp           load_alignments_wtclp         wtclp.c:111-180   (columns 0-4, 5-9, 11 of >= 12)
 * At BASELINE configs[2] that is 481 648 records / 3.4 GB with the CIGAR column, which the zmo pipeline throws away in the pipe
 * (`wtzmo -fo - | cut -f1-16`, smartdenovo.pl:58).  A binary stream carries the same integers in 64 bytes per record and names once.
 *
 * Stream layout (little endian, sequential - it can travel through a pipe):
 *     header      magic "WTZOVLB1", u32 version (1), u32 flags (0), u64 n_reads, u64 name_bytes
 *     name table  n_reads x { u32 read_length, u16 name_length, name bytes (no terminator) }   in the WRITER's read-id order; name_bytes = its size
 *     records     wtz_ovlb_rec_t x N until end of stream
 * A record holds the 16 text columns as integers: ids index the name table; `aln` is the denominator of the identity column
 * (`%0.3f` of mat / aln, wtzmo.c:1240 with aln == 0 -> 1, wtzmo.c:1220), so the text form can be reproduced byte for byte:
 * wtz_ovlb_format16() prints exactly what `cut -f1-16` of the text output holds, for consumers that stay on text.
 */
#ifndef WTZ_OVLB_H
#define WTZ_OVLB_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define WTZ_OVLB_MAGIC "WTZOVLB1"
#define WTZ_OVLB_VERSION 1u

typedef struct {
	uint32_t id1, id2;           /* query / candidate read (columns 1 and 6 by name-table index) */
	uint32_t aln;                /* alignment length behind the identity column (>= 1) */
	int32_t  tb, te;             /* columns 4, 5: begin / end on read 1 */
	int32_t  qb, qe;             /* columns 9, 10: begin / end on read 2, in the coordinates of its printed strand */
	int32_t  score;              /* column 11 */
	int32_t  mat, mis, ins, del; /* columns 13-16 */
	uint8_t  dir2;               /* column 7: 0 '+', 1 '-' (read 1 is always '+') */
	uint8_t  pad[3];
	uint32_t n_cigar;            /* reserved: CIGAR words that follow the record (0 in version 1: the zmo pipeline drops the column) */
	uint64_t reserved;
} wtz_ovlb_rec_t;                /* 64 bytes */

typedef struct { char magic[8]; uint32_t version, flags; uint64_t n_reads, name_bytes; } wtz_ovlb_hdr_t;      /* 32 bytes */

/* ---- writer ---- */
static inline int wtz_ovlb_write_header(FILE *fp, uint64_t n_reads, const char *const *names, const uint32_t *rdlen){
	wtz_ovlb_hdr_t h; memcpy(h.magic, WTZ_OVLB_MAGIC, 8); h.version = WTZ_OVLB_VERSION; h.flags = 0; h.n_reads = n_reads; h.name_bytes = 0;
	for(uint64_t i = 0; i < n_reads; i++){ size_t l = strlen(names[i]); if(l > 0xFFFFu) return -1; h.name_bytes += 6 + l; }
	if(fwrite(&h, sizeof h, 1, fp) != 1) return -1;
	for(uint64_t i = 0; i < n_reads; i++){
		const uint32_t L = rdlen[i]; const uint16_t l = (uint16_t)strlen(names[i]);
		if(fwrite(&L, 4, 1, fp) != 1 || fwrite(&l, 2, 1, fp) != 1 || (l && fwrite(names[i], 1, l, fp) != l)) return -1;
	}
	return 0;
}

/* ---- reader ---- */
typedef struct {
	FILE *fp; uint64_t n_reads; uint32_t flags;
	char *blob; char **names; uint32_t *rdlen;      /* names[i] is NUL-terminated inside blob */
} wtz_ovlb_reader_t;

/* 0 = fine, -1 = not a binary overlap stream / truncated.  `first8` (may be NULL): bytes the caller already consumed to sniff the format. */
static inline int wtz_ovlb_open(wtz_ovlb_reader_t *r, FILE *fp, const char *first8){
	wtz_ovlb_hdr_t h; memset(r, 0, sizeof *r);
	if(first8){ memcpy(&h, first8, 8); if(fread((char*)&h + 8, sizeof h - 8, 1, fp) != 1) return -1; }
	else if(fread(&h, sizeof h, 1, fp) != 1) return -1;
	if(memcmp(h.magic, WTZ_OVLB_MAGIC, 8) != 0 || h.version != WTZ_OVLB_VERSION) return -1;
	r->fp = fp; r->n_reads = h.n_reads; r->flags = h.flags;
	char *raw = (char*)malloc(h.name_bytes + 1);
	r->blob = (char*)malloc(h.name_bytes + h.n_reads + 1);          /* the names again, each with a terminator */
	r->names = (char**)malloc(sizeof(char*) * (h.n_reads + 1)); r->rdlen = (uint32_t*)malloc(4 * (h.n_reads + 1));
	if(!raw || !r->blob || !r->names || !r->rdlen){ free(raw); return -1; }
	if(h.name_bytes && fread(raw, 1, h.name_bytes, fp) != h.name_bytes){ free(raw); return -1; }
	uint64_t o = 0, w = 0;
	for(uint64_t i = 0; i < h.n_reads; i++){
		if(o + 6 > h.name_bytes){ free(raw); return -1; }
		uint32_t L; uint16_t l; memcpy(&L, raw + o, 4); memcpy(&l, raw + o + 4, 2); o += 6;
		if(o + l > h.name_bytes){ free(raw); return -1; }
		r->rdlen[i] = L; r->names[i] = r->blob + w; memcpy(r->blob + w, raw + o, l); r->blob[w + l] = 0; w += (uint64_t)l + 1; o += l;
	}
	free(raw);
	return 0;
}
/* 1 = a record, 0 = end of stream, -1 = truncated or ids out of range */
static inline int wtz_ovlb_next(wtz_ovlb_reader_t *r, wtz_ovlb_rec_t *rec){
	const size_t n = fread(rec, 1, sizeof *rec, r->fp);
	if(n == 0) return 0;
	if(n != sizeof *rec || rec->id1 >= r->n_reads || rec->id2 >= r->n_reads) return -1;
	for(uint32_t k = 0; k < rec->n_cigar; k++){ uint32_t w; if(fread(&w, 4, 1, r->fp) != 1) return -1; }      /* version 1 writers never set it */
	return 1;
}
static inline void wtz_ovlb_close(wtz_ovlb_reader_t *r){ free(r->blob); free(r->names); free(r->rdlen); memset(r, 0, sizeof *r); }

/* the identity column as the text writers print it (wtzmo.c:1240): "%0.3f" of mat / aln in double */
static inline int wtz_ovlb_identity_text(const wtz_ovlb_rec_t *rec, char *buf){ return sprintf(buf, "%0.3f", 1.0 * rec->mat / (rec->aln ? rec->aln : 1u)); }

/* the first 16 columns of the record's text line, tab separated, NO line end: byte-identical to `cut -f1-16` of the text output.
 * buf must hold both names + 200 bytes.  Returns the length. */
static inline size_t wtz_ovlb_format16(const wtz_ovlb_reader_t *r, const wtz_ovlb_rec_t *rec, char *buf){
	size_t k = (size_t)sprintf(buf, "%s\t+\t%u\t%d\t%d\t%s\t%c\t%u\t%d\t%d\t%d\t", r->names[rec->id1], r->rdlen[rec->id1], rec->tb, rec->te,
		r->names[rec->id2], "+-"[rec->dir2 & 1], r->rdlen[rec->id2], rec->qb, rec->qe, rec->score);
	k += (size_t)wtz_ovlb_identity_text(rec, buf + k);
	k += (size_t)sprintf(buf + k, "\t%d\t%d\t%d\t%d", rec->mat, rec->mis, rec->ins, rec->del);
	return k;
}

#endif
