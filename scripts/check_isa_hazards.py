#!/usr/bin/env python
"""Scan the gfx950 code of every HIP translation unit for one instruction pattern hipcc leaves unpadded and gfx950 was
seen to get wrong (round 3, the 128 x 160 tile GEMM's epilogue):

    buffer_store_dwordx4 v[0:3], v9, s[8:11], s2 offen      <- 16-byte store WITH an SGPR offset
    v_or_b32_e32 v2, s7, v10                                <- VALU rewrites a data register in the next slot

The store carried the NEW v2 (a row index) in a few percent of the launches: the data registers of a store wider than 8
bytes are fetched over more than one cycle.  LLVM's hazard table pads this only for stores WITHOUT an SGPR offset (the
documented case); the kernel was fixed by computing every offset first and issuing the stores back to back.

Flagged: a VMEM store of 12 or 16 bytes whose data registers are written by a VALU instruction in one of the next
`--slots` (default 2) instruction slots.  (Loads into the same registers are not flagged: their data comes back tens of cycles later.)

What it does NOT cover: stores whose overwrite sits behind a taken branch; the lab builds under scripts/lab/ (not part of the
library); any other unpadded hazard class.  The kernels with 16-byte SGPR-offset stores (gemm_m128_kernel, rowpanel_kernel) do
not rely on it alone: their epilogues compute every offset first and issue the stores back to back behind a sched_barrier.

Usage:  python scripts/check_isa_hazards.py            (after build(): reads adaface_amd/_build/*.o)
Exit code 1 if anything is flagged.  tests/test_host_cpu.py runs scan() in the CPU suite.
"""
import argparse
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
LLVM = Path("/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
WIDE_STORE = re.compile(r"^(buffer_store_dwordx[34]|global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\b")


def _regs(tok: str) -> set:
    tok = tok.strip()
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def disassemble(obj: Path, tmp: Path):
    """Device code of one host object as (kernel name, [instruction text]) pairs; None if the object has no device code."""
    fat, co = tmp / (obj.name + ".fat"), tmp / (obj.name + ".co")
    subprocess.run([LLVM / "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    if not fat.exists() or fat.stat().st_size == 0:
        return None
    r = subprocess.run([LLVM / "clang-offload-bundler", "--unbundle", "--type=o", f"--targets={TARGET}", f"--input={fat}",
                        f"--output={co}"], capture_output=True, text=True)
    if r.returncode != 0:
        return None
    text = subprocess.run([LLVM / "llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    funcs, cur = [], None
    for ln in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            cur = (m.group(1), [])
            funcs.append(cur)
        elif cur is not None and ln.startswith("\t"):
            cur[1].append(ln.split("//")[0].strip())
    return funcs


def scan_function(name: str, ins: list, slots: int):
    hits = []
    for i, t in enumerate(ins):
        if not WIDE_STORE.match(t):
            continue
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
        data = _regs(ops[0]) if t.startswith("buffer") else _regs(ops[1])
        for k in range(1, slots + 1):
            if i + k >= len(ins):
                break
            u = ins[i + k]
            if u.startswith(("s_nop", "s_barrier", "s_endpgm", "s_branch", "s_cbranch", "s_setpc", "s_swappc")):
                break                     # (s_nop N = N + 1 wait states, one is what the documented hazard asks for; control
                                          # flow: past the window this scan can judge)
            # (an s_waitcnt takes a slot but guarantees no wait state when its counters are already satisfied: scan on)
            if u.startswith("v_") and not u.startswith(("v_cmp", "v_nop")):
                dst = u.split(None, 1)[1].split(",")[0]
                if _regs(dst) & data:
                    hits.append((name, t, u, k))
    return hits


def scan(slots: int = 2):
    """-> (number of wide stores seen, list of flagged (kernel, store, overwriting instruction, slot))"""
    objs = sorted((ROOT / "adaface_amd" / "_build").glob("*.o"))
    if not objs:
        raise RuntimeError("no objects under adaface_amd/_build: run __graft_entry__.build() first")
    # the objects must be what the sources say: a stale object would be scanned in place of the code that ships
    csrc = ROOT / "adaface_amd" / "csrc"
    newest_h = max(h.stat().st_mtime for h in list(csrc.glob("*.h")) + [ROOT / "include" / "adaface_hip.h"])
    for obj in objs:
        src = csrc / obj.name[:-2]
        if src.exists() and obj.stat().st_mtime < max(src.stat().st_mtime, newest_h):
            raise RuntimeError(f"{obj.name} is older than its sources: rebuild (python -m adaface_amd.build) before scanning")
    stores, hits = 0, []
    with tempfile.TemporaryDirectory() as td:
        for obj in objs:
            funcs = disassemble(obj, Path(td))
            for name, ins in funcs or []:
                stores += sum(1 for t in ins if WIDE_STORE.match(t))
                hits += scan_function(name, ins, slots)
    return stores, hits


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=2)
    a = ap.parse_args()
    n, hits = scan(a.slots)
    print(f"{n} stores of 12/16 bytes scanned, {len(hits)} flagged")
    for h in hits:
        print("  ", *h, sep=" | ")
    sys.exit(1 if hits else 0)
