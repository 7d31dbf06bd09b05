#!/usr/bin/env python3
"""Per-section instruction budget of a search kernel's hop loop, from the ISA hipcc emits (VERDICT r4 "next" #1a).

usage: scripts/hop_budget.py <unit> <kernel-substring> [--all]
  unit: 1..5 = csrc/search_inst.hip with -DSEARCH_INST_SHAPE=unit (1 Shape2x4, 2 Shape4x2, 3 Shape8x2, 4 Shape12x2, 5 Shape2x2)
  kernel-substring: matched against the demangled name, e.g. "beam<0, Shape12x2, 4, true, false>"

The unit is compiled to assembly with -gline-tables-only (line tables do not change the code: the instruction count is compared with
a build without them), every instruction is attributed to the source line of its last .loc — inlined helpers keep their OWN lines, so
beam_next, tagset_test_and_set, score_rows ... separate by themselves — and lines are mapped to the sections of a hop
(device_search.h hnsw_search_kernel_beam: pop / stop test / team: publish + package look-up / link list / visited test / compaction +
job posting / scoring / accept loop / prune).  Only instructions between the hop loop's first and last instruction in layout order are
counted as "in the loop" (cold blocks the compiler moved behind the function's end are listed separately).  Counts are STATIC: one
per instruction in the binary.  An executed count needs trip counts (rows per hop, accepted rows per hop) — the table prints the
per-iteration bodies separately where the source has a loop (accept loop per row, scoring per pass) so that they can be weighted."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pg_embedding_amd", "csrc")


def src_lines(name):
    with open(os.path.join(CSRC, name)) as f:
        return f.read().splitlines()


def find_line(lines, needle, start=0):
    for i in range(start, len(lines)):
        if needle in lines[i]:
            return i + 1
    raise SystemExit(f"marker not found: {needle}")


def sections():
    """line ranges of device_search.h / device_dist.h -> section names (found by markers, so edits do not silently shift them)"""
    ds = src_lines("device_search.h")
    L = lambda s, a=0: find_line(ds, s, a)
    k0 = L("void hnsw_search_kernel_beam(const SearchArgs a)")
    hop0 = L("// hnswalg.cpp:67-112", k0)
    pop_end = L("if (HOP_STAMPS && a.team_dbg) { hs1 = hop_stamp(); hs_pop += hs1 - hs0; hs0 = hs1; }", hop0)
    link0 = L("for (uint32_t j0 = 0; j0 < a.maxM; j0 += 64)", hop0)
    vis0 = L("bool isnew = false;", link0)
    comp0 = L("const uint64_t mask = __ballot(isnew);", vis0)
    score0 = L("if (nscore)", comp0)
    acc0 = L("const uint32_t t_mine = newid[lane];", score0)
    hop_end = L("if (TEAM && lane == 0) ctl[wib].state = 0u;", acc0)
    emit0 = L("// ---- emit:", hop_end)
    kend = len(ds)
    fn = lambda name: (L(f"{name}("), )
    rng = []
    def helper(name, sec, endmarker="\n}"):
        a = L(name)
        b = a
        while not ds[b - 1].startswith("}"):
            b += 1
        rng.append((a - 2, b, sec))
    helper("__device__ __forceinline__ bool beam_next(", "pop")
    helper("__device__ __forceinline__ uint32_t wave_min_u32(", "pop")
    helper("__device__ __forceinline__ uint32_t beam_count_lt(", "stop test")
    helper("__device__ __forceinline__ uint32_t beam_count_le(", "accept")
    helper("__device__ __forceinline__ void beam_set(", "accept")
    helper("__device__ __forceinline__ uint32_t beam_select(", "prune")
    helper("__device__ __forceinline__ uint32_t beam_compact(", "prune")
    helper("__device__ __forceinline__ int tagset_test_and_set(", "visited")
    helper("__device__ __forceinline__ void tagset_split(", "visited")
    helper("__device__ __forceinline__ uint32_t tag_match(", "visited")
    helper("__device__ __forceinline__ uint64_t team_find(", "team look-up")
    helper("__device__ __forceinline__ TeamView team_view(", "team look-up")
    helper("__device__ __forceinline__ uint64_t uniform_u64(", "team look-up")
    helper("__device__ __forceinline__ uint32_t lc_slot(", "team look-up")
    helper("__device__ __forceinline__ uint32_t lane_rank(", "(lane_rank)")
    body = [(k0, hop0, "query set-up"), (hop0, pop_end + 1, "pop"), (pop_end + 1, link0, "team look-up"), (link0, vis0, "link list"),
            (vis0, comp0, "visited"), (comp0, score0, "compaction"), (score0, acc0, "scoring"), (acc0, hop_end, "accept"),
            (hop_end, emit0, "walk end"), (emit0, kend, "emit")]
    return rng, body, (hop0, hop_end)


CLASSES = [("mfma", r"v_mfma"), ("lane xchg", r"v_readlane|v_writelane|v_readfirstlane|ds_bpermute|ds_permute|v_permlane|_dpp|v_mov_b32_dpp"),
           ("scratch", r"scratch_"), ("vmem", r"global_|buffer_|flat_"), ("lds", r"ds_"), ("smem", r"s_load|s_buffer_load"),
           ("waitcnt", r"s_waitcnt"), ("branch", r"s_cbranch|s_branch|s_setpc|s_call"), ("nop/sleep", r"s_nop|s_sleep|s_barrier|s_sched"),
           ("valu", r"v_"), ("salu", r"s_")]


def classify(ins, text):
    if "_dpp" in text or "row_" in text or "quad_perm" in text:
        return "lane xchg"
    for name, pat in CLASSES:
        if re.match(pat, ins):
            return name
    return "other"


def main():
    unit, sub = sys.argv[1], sys.argv[2]
    extra = ["-DSEARCH_INST_SHAPE=" + unit]
    helpers, body, (hop0, hop_end) = sections()
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-gline-tables-only", *extra,
               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, "search_inst.hip")]
        subprocess.run(cmd, check=True, capture_output=True)
        text = open(out).read().splitlines()
    # file numbers
    fileno = {}
    for ln in text:
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', ln)
        if m:
            fileno[int(m.group(1))] = os.path.basename(m.group(2))
    # the kernel
    start = end = None
    want = None
    for i, ln in enumerate(text):
        m = re.match(r"^(_ZN5pgemb\w+):", ln)
        if m:
            dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            dem = dem.replace("pgemb::", "")
            if sub in dem:
                start, want = i, dem
        if start is not None and end is None and ln.startswith(".Lfunc_end") and i > start:
            end = i
    if start is None:
        raise SystemExit("kernel not found: " + sub)
    print(f"# {want}\n")

    def section_of(f, line):
        if f == "device_dist.h":
            return "scoring"
        if f != "device_search.h":
            return None                     # header intrinsics: keep the previous section
        for a, b, s in helpers:
            if a <= line <= b:
                return s
        for a, b, s in body:
            if a <= line < b:
                return s
        return "other"

    cur = "query set-up"
    cur_line = 0
    rows = []          # (pos, section, class, in_body_line)
    for i in range(start, end):
        ln = text[i].strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", ln)
        if m:
            s = section_of(fileno.get(int(m.group(1)), "?"), int(m.group(2)))
            if s == "(lane_rank)":
                s = None
            if s:
                cur = s
            if fileno.get(int(m.group(1))) == "device_search.h":
                cur_line = int(m.group(2))
            continue
        if not ln or ln.startswith((".", ";", "//")) or ln.endswith(":"):
            continue
        ins = ln.split()[0]
        rows.append((i, cur, classify(ins, ln), cur_line))
    # hop loop extent in layout order: first instruction whose body line is inside the hop loop .. last such
    inloop = [k for k, r in enumerate(rows) if hop0 <= r[3] < hop_end and r[1] not in ("query set-up", "emit", "walk end")]
    lo, hi = min(inloop), max(inloop)
    k0 = body[0][0]
    while lo > 0 and not (k0 <= rows[lo - 1][3] < hop0):      # the loop opens with inlined helpers (beam_next): their lines are not the body's
        lo -= 1
    secs = ["pop", "stop test", "team look-up", "link list", "visited", "compaction", "scoring", "accept", "prune", "other"]
    cols = ["valu", "salu", "lane xchg", "lds", "vmem", "smem", "scratch", "branch", "waitcnt", "nop/sleep", "mfma", "other"]
    table = {s: {c: 0 for c in cols} for s in secs}
    outside = {c: 0 for c in cols}
    for k, (pos, sec, cl, bl) in enumerate(rows):
        if lo <= k <= hi and sec in table:
            table[sec][cl] += 1
        else:
            outside[cl] += 1
    print("| section (static instructions inside the hop loop) | " + " | ".join(cols) + " | total |")
    print("|---|" + "---|" * (len(cols) + 1))
    tot = {c: 0 for c in cols}
    for s in secs:
        r = table[s]
        if sum(r.values()) == 0:
            continue
        print(f"| {s} | " + " | ".join(str(r[c]) for c in cols) + f" | {sum(r.values())} |")
        for c in cols:
            tot[c] += r[c]
    print("| **hop loop** | " + " | ".join(str(tot[c]) for c in cols) + f" | {sum(tot.values())} |")
    print("| outside (set-up, entry point, emit, clean-up, cold blocks) | " + " | ".join(str(outside[c]) for c in cols) + f" | {sum(outside.values())} |")


if __name__ == "__main__":
    main()
