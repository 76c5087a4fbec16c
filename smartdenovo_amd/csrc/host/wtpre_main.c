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

static int usage(void){
	printf(
	"WTPRE: Prepare raw reads for assembly\n"
	"SMARTdenovo: Ultra-fast de novo assembler for high noisy long reads\n"
	"Usage: wtpre [options] <raw_reads_file:fq/fa>\n"
	"Options:\n"
	" -o <string> Output of processed reads, [-]\n"
	" -f          Force overwrite output file\n"
	" -L          Keep all subreads in a well, default: the longest one\n"
	" -J <int>    Jack knife of read length, [0]\n"
	" -c <int>    Clip <-c> bases at both ends, [0]\n"
	" -p <string> Change the read name into {\"%%s%%012d\", <-p>}, [pb]\n"
	"\n"
	"Example: \n"
	"$> wtpre -J 5000 -p pb my_raw_reads_1.fq my_raw_reads_2.fq >wt.fa\n"
	"\n");
	return 1;
}

static void set_str(hx_str_t *d, const char *p, size_t n){ d->n = 0; hx_str_add(d, p, n); }

int main(int argc, char **argv){
	int longest = 1, min_len = 0, clp_len = 0, overwrite = 0, c;
	const char *prefix = "pb", *outf = NULL;
	while((c = getopt(argc, argv, "ho:fLJ:c:p:")) >= 0){
		switch(c){
			case 'h': return usage();
			case 'o': outf = optarg; break;
			case 'f': overwrite = 1; break;
			case 'L': longest = 0; break;
			case 'J': min_len = atoi(optarg); break;
			case 'c': clp_len = atoi(optarg); break;
			case 'p': prefix = optarg; break;
			default: return usage();
		}
	}
	if(optind == argc) return usage();
	if(!overwrite && outf && strcmp(outf, "-")){ FILE *t = fopen(outf, "r"); if(t){ fclose(t); fprintf(stderr, "File exists! '%s'\n\n", outf); return usage(); } }
	hx_reader_t *fr = hx_reader_open(argv + optind, argc - optind);
	if(!fr){ fprintf(stderr, " -- Cannot open %s --\n", argv[optind]); return 1; }
	FILE *out = outf ? (strcmp(outf, "-") == 0 ? stdout : fopen(outf, "w")) : stdout;
	if(!out){ fprintf(stderr, " -- Cannot open %s for write --\n", outf); return 1; }
	hx_str_t tag = {0, 0, 0}, dsc = {0, 0, 0}, seq = {0, 0, 0};
	hx_str_t w_tag = {0, 0, 0}, w_dsc = {0, 0, 0}, w_seq = {0, 0, 0};       /* the well's best subread so far */
	hx_str_add(&w_tag, "", 0); hx_str_add(&w_dsc, "", 0); hx_str_add(&w_seq, "", 0);
	unsigned long long idx = 0; int max = 0;
	while(hx_reader_seq_desc(fr, &tag, &dsc, &seq)){
		if(!tag.s) hx_str_add(&tag, "", 0);
		if(!dsc.s) hx_str_add(&dsc, "", 0);
		if(!seq.s) hx_str_add(&seq, "", 0);
		const int seqlen = (int)seq.n - 2 * clp_len;
		if(seqlen < min_len) continue;
		char *seqstr = seq.s + clp_len;
		seqstr[seqlen] = 0;
		if(longest){
			int size = (int)tag.n, f = 0;
			while(size){
				const char ch = tag.s[size - 1];
				if(ch <= '9' && ch >= '0') size--;
				else if(ch == '_'){ if(f) break; size--; f = 1; }
				else if(ch == '/'){ if(f == 1){ size--; f = 2; } break; }
				else break;
			}
			if(size <= 0 || f < 2) size = (int)tag.n;
			if((int)w_tag.n == size && strncmp(w_tag.s, tag.s, (size_t)size) == 0){
				if(seqlen > max){ set_str(&w_tag, tag.s, (size_t)size); set_str(&w_dsc, dsc.s, dsc.n); set_str(&w_seq, seqstr, (size_t)seqlen); max = (int)seq.n; }
			} else {
				if(w_tag.n) fprintf(out, ">%s%012llu%s\n%s\n", prefix, idx++, w_dsc.s, w_seq.s);
				set_str(&w_tag, tag.s, (size_t)size); set_str(&w_dsc, dsc.s, dsc.n); set_str(&w_seq, seqstr, (size_t)seqlen); max = (int)seq.n;
			}
		} else if(seqlen >= min_len){
			fprintf(out, ">%s%012llu%s\n%s\n", prefix, idx++, dsc.s, seqstr);
		}
	}
	if(w_tag.n) fprintf(out, ">%s%012llu%s\n%s\n", prefix, idx++, w_dsc.s, w_seq.s);
	hx_reader_close(fr);
	if(out != stdout) fclose(out);
	return 0;
}
