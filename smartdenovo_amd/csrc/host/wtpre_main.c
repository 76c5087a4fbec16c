/*
 * wtpre_main.c — drop-in `wtpre` (smartdenovo.pl:43-44: `wtpre -J <min_len> reads | gzip -c -1 > prefix.fa.gz`), the step in
 * front of the overlapper: renames reads to <prefix><12-digit serial>, keeps the longest subread of a PacBio well, drops short reads, clips.
 * Pure text in / text out on the host — the restatement follows wtpre.c:44-137 statement by statement because every branch is visible in
 * the output; the other half of f4, the FASTA -> 2-bit packing of the reads it writes, runs on the device (wtz_upload_reads_ascii).
 *
 * Kept quirks: the well name is the tag minus a trailing `/<digits>_<digits>` (wtpre.c:89-108), anything else leaves the tag whole; a later
 * subread of the same well replaces the stored one when its CLIPPED length exceeds the stored read's UNCLIPPED length (wtpre.c:110,114);
 * -J filters on the clipped length; usage goes to stdout with return 1.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "wtz_host.h"

static int usage(void){      /* to stdout, return 1, like wtpre.c:22-42 */
	fputs(
	"WTPRE (MI355X tool set): prepare raw reads for assembly - rename, longest subread per well, length filter, clipping\n"
	"Usage: wtpre [options] <raw_reads_file:fq/fa> [more files]\n"
	" -o <file>   output [-]            -f  overwrite an existing output\n"
	" -L          keep every subread of a well (default: the longest one)\n"
	" -J <int>    drop reads shorter than this after clipping [0]\n"
	" -c <int>    clip this many bases at both ends [0]\n"
	" -p <string> reads are renamed <-p><12-digit serial> [pb]\n"
	"e.g.  wtpre -J 5000 -p pb raw_1.fq raw_2.fq > wt.fa\n", stdout);
	return 1;
}

/* length of the WELL part of a subread name `<well>/<digits>_<digits>` (either digit string may be empty); the whole name when it does not end
 * like that or nothing would be left in front (wtpre.c:89-108) */
static size_t well_name_len(const char *tag, size_t n){
	size_t k = n;
	while(k && tag[k - 1] >= '0' && tag[k - 1] <= '9') k--;
	if(k == 0 || tag[k - 1] != '_') return n;
	k--;
	while(k && tag[k - 1] >= '0' && tag[k - 1] <= '9') k--;
	if(k == 0 || tag[k - 1] != '/') return n;
	k--;
	return k ? k : n;
}

typedef struct { hx_str_t well, desc, seq; int raw_len; } kept_t;       /* the subread kept for the current well */
static void keep(kept_t *b, const char *tag, size_t wn, const hx_str_t *desc, const char *seq, int seqlen, int raw_len){
	b->well.n = 0; hx_str_add(&b->well, tag, wn);
	b->desc.n = 0; hx_str_add(&b->desc, desc->s ? desc->s : "", desc->n);
	b->seq.n = 0; hx_str_add(&b->seq, seq, (size_t)seqlen);
	b->raw_len = raw_len;
}
static void emit(FILE *out, const char *prefix, unsigned long long *serial, const char *desc, const char *seq){
	fprintf(out, ">%s%012llu%s\n%s\n", prefix, (*serial)++, desc, seq);
}

int main(int argc, char **argv){
	int per_well = 1, min_len = 0, clip = 0, overwrite = 0, c;
	const char *prefix = "pb", *outf = NULL;
	while((c = getopt(argc, argv, "ho:fLJ:c:p:")) >= 0){
		if(c == 'o') outf = optarg;
		else if(c == 'f') overwrite = 1;
		else if(c == 'L') per_well = 0;
		else if(c == 'J') min_len = atoi(optarg);
		else if(c == 'c') clip = atoi(optarg);
		else if(c == 'p') prefix = optarg;
		else return usage();                     /* -h and anything unknown */
	}
	if(optind == argc) return usage();
	const int to_stdout = (outf == NULL || strcmp(outf, "-") == 0);
	if(!to_stdout && !overwrite && access(outf, F_OK) == 0){ fprintf(stderr, "File exists! '%s'\n\n", outf); return usage(); }
	hx_reader_t *fr = hx_reader_open(argv + optind, argc - optind);
	if(!fr){ fprintf(stderr, " -- Cannot open %s --\n", argv[optind]); return 1; }
	FILE *out = to_stdout ? stdout : fopen(outf, "w");
	if(!out){ fprintf(stderr, " -- Cannot open %s for write --\n", outf); return 1; }
	hx_str_t tag = {0, 0, 0}, dsc = {0, 0, 0}, seq = {0, 0, 0};
	kept_t best; memset(&best, 0, sizeof best);
	hx_str_add(&best.well, "", 0); hx_str_add(&best.desc, "", 0); hx_str_add(&best.seq, "", 0);
	unsigned long long serial = 0;
	while(hx_reader_seq_desc(fr, &tag, &dsc, &seq)){
		if(!tag.s) hx_str_add(&tag, "", 0);
		if(!dsc.s) hx_str_add(&dsc, "", 0);
		if(!seq.s) hx_str_add(&seq, "", 0);
		const int seqlen = (int)seq.n - 2 * clip;      /* -J applies to the clipped length (wtpre.c:84-85) */
		if(seqlen < min_len) continue;
		char *body = seq.s + clip;
		body[seqlen] = 0;
		if(!per_well){ emit(out, prefix, &serial, dsc.s, body); continue; }
		const size_t wn = well_name_len(tag.s, tag.n);
		const int same_well = (best.well.n == wn && strncmp(best.well.s, tag.s, wn) == 0);
		if(same_well){
			/* the reference compares the newcomer's CLIPPED length with the kept read's UNCLIPPED one (wtpre.c:110,114) */
			if(seqlen > best.raw_len) keep(&best, tag.s, wn, &dsc, body, seqlen, (int)seq.n);
		} else {
			if(best.well.n) emit(out, prefix, &serial, best.desc.s, best.seq.s);
			keep(&best, tag.s, wn, &dsc, body, seqlen, (int)seq.n);
		}
	}
	if(best.well.n) emit(out, prefix, &serial, best.desc.s, best.seq.s);
	hx_reader_close(fr);
	if(out != stdout) fclose(out);
	return 0;
}
