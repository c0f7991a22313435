#!/usr/bin/env python
"""Build-time audit of vh_gemm_ws.hip's emitted ISA (run by __graft_entry__.build()).

The kernel keeps its weight ring in v[192:255], named literally inside inline-asm statements, and relies on hipcc never
allocating those registers itself (vh_gemm_ws.hip: ws_ldw).  This script reads the gfx950 assembly (`-save-temps`) and
fails unless, for every k_gemm_ws kernel:
  * no instruction OUTSIDE an `;;#ASMSTART ... ;;#ASMEND` block names a VGPR >= RING0 (single registers or ranges),
  * at most a handful of scratch instructions exist (spilled per-tile scalars; accumulators in scratch fail the build),
  * the kernel descriptor allocates all 256 VGPRs.

  python profiles/audit_ws.py <file.s> [--ring0 192]
"""
import re
import sys


def audit(path, ring0=192, max_scratch=64):
    txt = open(path).read().splitlines()
    errs, in_asm, kern, seen, scratch = [], False, None, 0, []
    reg1 = re.compile(r"\bv(\d+)\b")
    regr = re.compile(r"\bv\[(\d+):(\d+)\]")
    meta = {}
    for ln, line in enumerate(txt, 1):
        st = line.strip()
        m = re.match(r"^(_Z\w*k_gemm_ws\w*):", st)
        if m:
            kern = m.group(1)
            seen += 1
        if st.startswith(".amdhsa_kernel"):
            kern = None if "k_gemm_ws" not in st else st.split()[-1]
        for key in (".amdhsa_next_free_vgpr", ".amdhsa_private_segment_fixed_size"):
            if st.startswith(key) and kern and "k_gemm_ws" in kern:
                meta[key] = int(st.split()[-1])
        if st.startswith(".end_amdhsa_kernel"):
            kern = None
        if ";;#ASMSTART" in st:
            in_asm = True
            continue
        if ";;#ASMEND" in st:
            in_asm = False
            continue
        if in_asm or not kern or st.startswith((";", ".", "//")) or st.endswith(":"):
            continue
        code = st.split(";")[0]
        hi = [int(x) for x in reg1.findall(code)] + [int(b) for a, b in regr.findall(code)]
        if any(r >= ring0 for r in hi):
            errs.append(f"{path}:{ln}: compiler instruction touches the weight ring: {code.strip()}")
        if "scratch_" in code:
            scratch.append(f"{path}:{ln}: {code.strip()}")
    if not seen:
        errs.append("no k_gemm_ws kernel found in " + path)
    # compiler spills are correct (its own registers) but slow: a handful of per-tile scalars is tolerated, accumulators in
    # scratch (hundreds of accesses: an epilogue loop left rolled, a row-tile count that no longer fits) are not
    if len(scratch) > max_scratch:
        errs += [f"{len(scratch)} scratch accesses (limit {max_scratch}), e.g. {scratch[0]}"]
    elif scratch:
        print(f"audit_ws: note: {len(scratch)} scratch accesses (spilled per-tile scalars), e.g. {scratch[0]}")
    if meta.get(".amdhsa_next_free_vgpr", 0) != 256:
        errs.append(f"kernel descriptor allocates {meta.get('.amdhsa_next_free_vgpr')} VGPRs, expected 256")
    # (a private segment WITHOUT scratch instructions is only reserved stack for SGPR spills that were lowered to VGPR
    # lanes: reported, not an error)
    if meta.get(".amdhsa_private_segment_fixed_size", 0) != 0:
        print(f"audit_ws: note: private segment of {meta.get('.amdhsa_private_segment_fixed_size')} bytes per lane, no scratch instruction")
    return errs


if __name__ == "__main__":
    ring0 = int(sys.argv[sys.argv.index("--ring0") + 1]) if "--ring0" in sys.argv else 192
    e = audit(sys.argv[1], ring0)
    for x in e[:40]:
        print(x)
    print(f"audit_ws: {'FAILED, ' + str(len(e)) + ' findings' if e else 'ok'}")
    sys.exit(1 if e else 0)
