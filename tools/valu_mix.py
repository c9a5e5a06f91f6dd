"""Static VALU instruction-class mix of the compositing kernels, from the device assembly hipcc emits with the Makefile's flags, and the
issue time it implies with the per-class costs tools/valu_bench.hip measured on the MI355X (profiles/r06_valu_microbenchmark.txt:
8 waves per SIMD, ns per wave-instruction per SIMD): plain f32 / integer 1.2, DPP row operations 1.76, packed f32 / f64 / f32<->f64
conversions 1.8, transcendentals (v_exp / v_rcp / v_log / v_sqrt / v_rsq) 3.4.

    python tools/valu_mix.py > profiles/valu_model.json        (runs on CPU: hipcc cross-compiles gfx950)

bench.py multiplies a kernel's measured SQ_INSTS_VALU (PMC pass) by the mix-weighted cost: the time the kernel's vector instructions
need on the chip's 1 024 SIMDs if nothing else ever stalled -- the floor the duration is compared with (`roofline.valu`).  The mix is
static (every instruction of the kernel counted once): the walk loop dominates both the static and the dynamic count of these kernels.
The file records the kernel-source hash it was made from (tools/srchash.py); bench.py ignores a model made from other sources."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gaussian-mesh-splatting_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import srchash  # noqa: E402

NS = {"plain": 1.2, "dpp": 1.76, "wide": 1.8, "trans": 3.4}          # tools/valu_bench.hip, 8 waves per SIMD
TRANS = ("v_exp_", "v_rcp_", "v_log_", "v_sqrt_", "v_rsq_", "v_sin_", "v_cos_")
KERNELS = {          # bench.py's kernel name -> (file, regex on the demangled name of the shipped instantiation)
    "blend_bwd": ("blend_micro.hip", r"micro_bwd_kernel<false, 2, 0, false, true, 256>"),
    "blend_head": ("blend_micro.hip", r"micro_head_kernel<4>"),
    "blend_fwd": ("blend_micro.hip", r"micro_fwd_kernel<4>"),
}


def classify(op):
    if op.endswith("_dpp") or "_dpp " in op:
        return "dpp"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_pk_") or "_f64" in op:
        return "wide"
    return "plain"


def device_asm(src, tmp):
    mk = open(os.path.join(CSRC, "Makefile")).read()
    m = re.search(r"^NOSLP\s*:=\s*(.*)$", mk, re.M)
    noslp = ["-fno-slp-vectorize"] if m and src in m.group(1).split() else []
    out = os.path.join(tmp, src + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function"] + noslp +
                   ["-S", "--cuda-device-only", src, "-o", out], cwd=CSRC, check=True, capture_output=True)
    return open(out).read()


def main():
    model = {"_source_hash": srchash.kernel_source_hash(), "ns_per_wave_instruction_per_simd": NS,
             "_how": "tools/valu_mix.py (static mix of the compiler's gfx950 assembly) x tools/valu_bench.hip (measured issue costs)", "kernels": {}}
    with tempfile.TemporaryDirectory() as tmp:
        asm = {}
        for name, (src, pat) in KERNELS.items():
            if src not in asm:
                asm[src] = device_asm(src, tmp)
            syms = re.findall(r"^(_Z\w+):", asm[src], re.M)
            dem = subprocess.run(["c++filt"], input="\n".join(syms), capture_output=True, text=True).stdout.split("\n")
            hit = [s for s, d in zip(syms, dem) if re.search(pat, d.replace("gms::", ""))]
            if not hit:
                print(f"{name}: no symbol matches {pat}", file=sys.stderr)
                continue
            body = asm[src].split(hit[0] + ":", 1)[1].split(".Lfunc_end", 1)[0].split("\n")

            def mix(lines):
                counts = {k: 0 for k in NS}
                for line in lines:
                    m = re.match(r"\s+(v_\w+)", line)
                    if not m:
                        continue
                    # DPP shows as a modifier on the operand list as well (row_ror / quad_perm / row_half_mirror ...)
                    op = m.group(1) + ("_dpp" if re.search(r"row_|quad_perm|row_mask", line) and not m.group(1).endswith("_dpp") else "")
                    counts[classify(op)] += 1
                tot = max(sum(counts.values()), 1)
                return {"static_valu": tot, "mix": {k: round(v / tot, 4) for k, v in counts.items()},
                        "ns_per_inst": round(sum(NS[k] * v for k, v in counts.items()) / tot, 4)}

            # the walk loop: from the loop header in front of the first v_exp_f32 to the last table add (backward) / the last
            # v_exp_f32's loop-closing branch (forward): where ~3/4 of the kernel's dynamic VALU instructions are issued
            exps = [i for i, l in enumerate(body) if "v_exp_f32" in l]
            heads = [i for i, l in enumerate(body) if "Loop Header" in l and i < exps[0]]
            adds = [i for i, l in enumerate(body) if "ds_add_u64" in l]
            end = adds[-1] if adds else next(i for i in range(exps[-1], len(body)) if re.match(r"\s+s_cbranch", body[i]) and i > exps[-1] + 20)
            whole, loop = mix(body), mix(body[heads[-1]:end + 1])
            w = 0.75          # share of the dynamic instructions the walk loop issues (tools/micro_phases.py trip counts x loop length)
            model["kernels"][name] = {"symbol": hit[0], "whole_kernel": whole, "walk_loop": loop, "walk_loop_dynamic_share": w,
                                      "ns_per_inst_mix_weighted": round(w * loop["ns_per_inst"] + (1 - w) * whole["ns_per_inst"], 4)}
    json.dump(model, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
