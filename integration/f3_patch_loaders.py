#!/usr/bin/env python3
"""f3: the reference's own overlap loaders reading include/wtz_ovlb.h streams - built and used by tests/test_f3_reference_loaders.py.

Nothing of the reference is committed here.  This script copies the reference's C files to a scratch directory (default /tmp/wtz_f3), inserts
THREE lines into wtlay.h and THREE into wtclp.c (located by the function names / statements they sit next to), corrects the out-of-bounds read of
wtclp.c:171-172, and builds with plain gcc - the recipe of oracle/Makefile, not the reference's build system:

    wtlay_bin   wtlay with   parse_overlap_item_strgraph  (wtlay.h:238-268) taking records from binary files   (also what the reference's wtgbo loads with: wtgbo_bin)
    wtclp_bin   wtclp with   load_alignments_wtclp        (wtclp.c:111-180) doing the same, and with the loop at wtclp.c:170-172 testing `i == size` BEFORE it
                reads ptrs[i] (the unpatched order reads one element past the initialised part of the array and indexes `hits` with it: the crash SURVEY measured)
    wtclp_fix   wtclp with only that correction (the text-path baseline the binary path is compared with)

The glue the inserted lines call is integration/wtz_ovlb_loaders.h (written for this repo against the reference's data structures).
usage: python integration/f3_patch_loaders.py [--ref /root/reference] [--out /tmp/wtz_f3]      -> prints the directory with the three programs
"""
import argparse, glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-w", "-O3", "-D_FILE_OFFSET_BITS=64", "-D_GNU_SOURCE", "-mpopcnt", "-mssse3"]


def insert_before(text, anchor, new, what):
    k = text.find(anchor)
    if k < 0:
        sys.exit("f3_patch_loaders: cannot find %s" % what)
    return text[:k] + new + text[k:]


def insert_after(text, start_anchor, anchor, new, what):
    s = text.find(start_anchor)
    k = text.find(anchor, s) if s >= 0 else -1
    if k < 0:
        sys.exit("f3_patch_loaders: cannot find %s" % what)
    k += len(anchor)
    return text[:k] + new + text[k:]


def patch_wtlay_h(t):
    fn = "int parse_overlap_item_strgraph("
    t = insert_before(t, fn, '#define WTZ_OVLB_FOR_WTLAY\n#include "wtz_ovlb_loaders.h"\n', "parse_overlap_item_strgraph in wtlay.h")
    return insert_after(t, fn, "int n, pb1, pb2;\n", "\tif(wtz_lay_binary_item(g, fr, dat)) return 1;\n", "the declarations of parse_overlap_item_strgraph")


def fix_wtclp_bounds(t):
    # `if((sids[1] = <hits[ptrs[i]]>) != sids[0] || i == size){`  ->  `if(i == size || (sids[1] = <...>) != sids[0]){`
    tail = " != sids[0] || i == wt->ptrs->size){"
    k = t.find(tail)
    s = t.rfind("if((sids[1] = ", 0, k) if k >= 0 else -1
    if s < 0:
        sys.exit("f3_patch_loaders: cannot find the loop condition of wtclp.c:171")
    inner = t[s + len("if("):k]
    return t[:s] + "if(i == wt->ptrs->size || " + inner + " != sids[0]){" + t[k + len(tail):]


def patch_wtclp_c(t, binary=True):
    t = fix_wtclp_bounds(t)
    if binary:
        fn = "void load_alignments_wtclp("
        t = insert_before(t, fn, '#define WTZ_OVLB_FOR_WTCLP\n#include "wtz_ovlb_loaders.h"\n', "load_alignments_wtclp in wtclp.c")
        t = insert_after(t, fn, "beg_counter(num);\n", "\twtz_clp_load_binary(wt, min_sm, fr);\n", "beg_counter in load_alignments_wtclp")
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference"); ap.add_argument("--out", default="/tmp/wtz_f3")
    a = ap.parse_args()
    if not os.path.isdir(a.ref):
        sys.exit("f3_patch_loaders: no reference sources under %s (build container only)" % a.ref)
    src = os.path.join(a.out, "src"); shutil.rmtree(a.out, ignore_errors=True); os.makedirs(src)
    for f in glob.glob(os.path.join(a.ref, "*.[ch]")):
        shutil.copy(f, src)
    open(os.path.join(src, "wtlay.h"), "w").write(patch_wtlay_h(open(os.path.join(a.ref, "wtlay.h")).read()))
    open(os.path.join(src, "wtclp_bin.c"), "w").write(patch_wtclp_c(open(os.path.join(a.ref, "wtclp.c")).read(), True))
    open(os.path.join(src, "wtclp_fix.c"), "w").write(patch_wtclp_c(open(os.path.join(a.ref, "wtclp.c")).read(), False))
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "integration")]
    for exe, main_c in (("wtlay_bin", "wtlay.c"), ("wtgbo_bin", "wtgbo.c"), ("wtclp_bin", "wtclp_bin.c"), ("wtclp_fix", "wtclp_fix.c")):
        subprocess.run(["gcc"] + FLAGS + inc + ["-o", os.path.join(a.out, exe), "file_reader.c", "ksw.c", main_c, "-lm", "-lpthread"], check=True, cwd=src)
    print(a.out)


if __name__ == "__main__":
    main()
