"""bench.py names the kernels a run executes by a hash of the CODE sections of the gfx950 code objects (bench.kernel_source_id): the PMC summaries under profiles/ carry
the id they were measured on, and the bench line says whether that is the id of the running build.  The id must not move when nothing a kernel is made of has moved -
two builds of the same sources differ in the order of the code object's symbol tables - and must move when the code does.  CPU only: works on the bytes of the built library."""
import os
import shutil
import struct

import pytest

import bench


def _sections(elf):
    shoff, = struct.unpack_from("<Q", elf, 0x28); shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    sec = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
    stro = sec[shstrndx][4]
    return {elf[stro + name:elf.index(b"\0", stro + name)]: (off, size) for name, _t, _f, _a, off, size, *_ in sec}


def _code_object_section(lib_bytes, want):
    """(absolute offset, size) of section `want` of the first amdgcn code object inside the library's .hip_fatbin"""
    foff, fsize = _sections(lib_bytes)[b".hip_fatbin"]
    fb = lib_bytes[foff:foff + fsize]
    assert fb[:24] == b"__CLANG_OFFLOAD_BUNDLE__"
    n, = struct.unpack_from("<Q", fb, 24); p = 32
    for _ in range(n):
        eoff, esize, tl = struct.unpack_from("<QQQ", fb, p); p += 24; triple = fb[p:p + tl]; p += tl
        if esize and b"amdgcn" in triple:
            o, z = _sections(fb[eoff:eoff + esize])[want]
            return foff + eoff + o, z
    raise AssertionError("no amdgcn code object in the library")


@pytest.fixture()
def lib_copy(tmp_path, monkeypatch):
    src = os.path.join(bench.ROOT, "smartdenovo_amd", "libwtzmo_hip.so")
    if not os.path.exists(src):
        pytest.skip("the device library is not built")
    root = tmp_path / "root"; (root / "smartdenovo_amd").mkdir(parents=True)
    dst = str(root / "smartdenovo_amd" / "libwtzmo_hip.so")
    shutil.copy(src, dst)
    ref = bench.kernel_source_id()
    monkeypatch.setattr(bench, "ROOT", str(root))
    return dst, ref


def test_kernel_id_is_a_property_of_the_code_not_of_the_symbol_tables(lib_copy):
    dst, ref = lib_copy
    assert len(ref) == 16 and ref != "unbuilt" and bench.kernel_source_id() == ref
    b = bytearray(open(dst, "rb").read())
    for sec in (b".dynstr", b".strtab"):          # what differs between two builds of the same sources (measured at the end of round 5)
        o, z = _code_object_section(bytes(b), sec)
        b[o + z // 2] ^= 0x20
    open(dst, "wb").write(bytes(b))
    assert bench.kernel_source_id() == ref


@pytest.mark.parametrize("sec", [b".text", b".rodata", b".note"])
def test_kernel_id_moves_with_the_code(sec, lib_copy):
    dst, ref = lib_copy
    b = bytearray(open(dst, "rb").read())
    o, z = _code_object_section(bytes(b), sec)
    b[o + z // 2] ^= 0x01
    open(dst, "wb").write(bytes(b))
    assert bench.kernel_source_id() != ref
