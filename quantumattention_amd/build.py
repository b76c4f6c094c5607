"""Ahead-of-time build of libqattn_hip.so for gfx950 with hipcc (no JIT on the hot path).

The reference JIT-compiles its CUDA kernel with nvcc on first call (src/quantum_attn/tk/attention.py:651-708);
here the library is built in-tree once (`python -m quantumattention_amd.build` or `__graft_entry__.build()`),
and the built `.so` travels with the source tree.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "_build")
# `-save-temps` output (79 MB of .s / .bc / .hipi for tests/test_kernel_resources.py and tools/kernel_resources.py) lives in a
# directory of its own that neither git nor gpurun ships (.gitignore, .gpurunignore); the objects of the shipped library do not
BUILD_TEMPS = os.path.join(HERE, "_build_temps")
LIB = os.path.join(HERE, "libqattn_hip.so")
# A/B baselines and tuning variants (`--variant=`) live in tools/ab_libs/, not beside the product library
AB_LIBS = os.path.join(os.path.dirname(HERE), "tools", "ab_libs")
SOURCES = ["qattn_quant.hip", "qattn_attn_v2.hip", "qattn_attn_v4.hip", "qattn_attn16.hip", "qattn_api.hip", "qattn_probe.hip", "qattn_attn_pv16.hip"]
# (source, extra flags, object name): the two big kernel files are compiled once per operand format / head dimension so that
# the build runs in parallel (the longest single translation unit sets the wall time)
# (a unit with a define is compiled through a two-line wrapper file named after the unit, so that -save-temps leaves one .s
# per unit)
UNITS = [
    ("qattn_attn_v2.hip", ["QATTN_ONLY_FMT 0", "QATTN_ONLY_IN16 2", "QATTN_STRIDED16 0"], "qattn_attn_v2_e4m3"),
    ("qattn_attn_v2.hip", ["QATTN_ONLY_FMT 1", "QATTN_ONLY_IN16 2", "QATTN_STRIDED16 0"], "qattn_attn_v2_e5m2"),
    ("qattn_attn_v2.hip", ["QATTN_ONLY_FMT 0", "QATTN_ONLY_IN16 3", "QATTN_STRIDED16 0"], "qattn_attn_v2_e4m3_f16"),   # the fused step from fp16 inputs
    ("qattn_attn_v2.hip", ["QATTN_ONLY_FMT 1", "QATTN_ONLY_IN16 3", "QATTN_STRIDED16 0"], "qattn_attn_v2_e5m2_f16"),
    # the hand-scheduled kernel once more for calls on strided views of q / v / out (fused step only): its dense units above keep
    # compile-time row sizes -- and with them the exact schedule they had before strides existed (csrc/qattn_attn.h QATTN_STRIDED16)
    ("qattn_attn_v2.hip", ["QATTN_ONLY_FMT 0", "QATTN_ONLY_IN16 2", "QATTN_V2_SV 1"], "qattn_attn_v2_e4m3_sv"),
    ("qattn_attn_v2.hip", ["QATTN_ONLY_FMT 1", "QATTN_ONLY_IN16 2", "QATTN_V2_SV 1"], "qattn_attn_v2_e5m2_sv"),
    ("qattn_attn_v2.hip", ["QATTN_ONLY_FMT 0", "QATTN_ONLY_IN16 3", "QATTN_V2_SV 1"], "qattn_attn_v2_e4m3_f16_sv"),
    ("qattn_attn_v2.hip", ["QATTN_ONLY_FMT 1", "QATTN_ONLY_IN16 3", "QATTN_V2_SV 1"], "qattn_attn_v2_e5m2_f16_sv"),
    ("qattn_attn_v4.hip", ["QATTN_ONLY_D 64"], "qattn_attn_v4_d64"),
    ("qattn_attn_v4.hip", ["QATTN_ONLY_D 128"], "qattn_attn_v4_d128"),
    ("qattn_attn_v4.hip", ["QATTN_ONLY_D 256"], "qattn_attn_v4_d256"),
    ("qattn_attn16.hip", [], "qattn_attn16"),
    ("qattn_quant.hip", [], "qattn_quant"),
    ("qattn_api.hip", [], "qattn_api"),
    ("qattn_probe.hip", [], "qattn_probe"),
    ("qattn_attn_pv16.hip", ["QATTN_STRIDED16 0"], "qattn_attn_pv16"),
    ("qattn_attn_pv16.hip", ["QATTN_PV16_SV 1"], "qattn_attn_pv16_sv"),   # the 16-bit-V launches on strided views (same reason as the _sv units above)
]
# (Until round 5 `--dev` built a second library with -DQATTN_DEV: timing-only ablation instantiations, per-segment cycle stamps, work logs and
# QATTN_* environment switches.  Round 6 took that scaffolding out of the sources -- identical product ISA, profiles/r06/isa_identity_*.log;
# the measurements it produced are in profiles/r01 .. r05 and docs/.  What remains for experiments: `--variant=<name> -DKNOB=value`.)
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-ffp-contract=off", "-Wall", "-Wno-unused-command-line-argument", "-Wno-unused-value", "-Wno-pass-failed"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the gfx950 library cannot be built")


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, save_temps: bool = False, verbose: bool = False, variant: str = "", extra_defines=()) -> str:
    """Compile what is stale and link the library; returns its path.  save_temps: compile every unit with -save-temps into
    BUILD_TEMPS instead (same flags, same code; the library is left alone) and return that directory.
    variant / extra_defines (development): a library of its own, tools/ab_libs/libqattn_<variant>.so, compiled with the given -D macros
    (kernel tuning knobs) for A/B runs with tools/ab.py or tools/kstats.sh."""
    extra_defines = sorted(extra_defines)
    if save_temps and (variant or extra_defines):
        # BUILD_TEMPS is what the resource / hazard tests read: only the product configuration may write there
        raise ValueError("save_temps is for the product configuration only (no variant, no -D macros)")
    if extra_defines and not variant:
        raise ValueError("-D macros need --variant=<name>: the product library is built without tuning macros")
    # the macros are part of the object directory's name: another set of -D values never reuses stale objects
    tag = ("_" + hashlib.sha1(" ".join(extra_defines).encode()).hexdigest()[:8]) if extra_defines else ""
    build_dir = BUILD_TEMPS if save_temps else BUILD + (f"_var_{variant}{tag}" if variant else "")
    lib = LIB
    if variant:
        os.makedirs(AB_LIBS, exist_ok=True)
        lib = os.path.join(AB_LIBS, f"libqattn_{variant}.so")
    os.makedirs(build_dir, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(os.path.dirname(HERE), "include", h) for h in ("qattn.h", "qattn_measure.h")]
    objs, jobs = [], []
    for src, defines, name in UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(build_dir, name + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers) or (save_temps and not os.path.exists(os.path.join(build_dir, name + "-hip-amdgcn-amd-amdhsa-gfx950.s"))):
            unit = s
            if defines:
                unit = os.path.join(build_dir, name + ".hip")
                text = "".join(f"#define {d}\n" for d in defines) + f'#include "{s}"\n'
                if not os.path.exists(unit) or open(unit).read() != text:
                    open(unit, "w").write(text)
            cmd = [hipcc] + FLAGS + [f"-D{d}" for d in extra_defines] + ["-c", unit, "-o", o]
            if save_temps:
                cmd += ["-save-temps=obj"]
            jobs.append(cmd)
    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=build_dir)
    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    if save_temps:
        return build_dir
    if force or jobs or _stale(lib, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    var = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--variant=")), "")
    print(build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, verbose=True, variant=var, extra_defines=[a[2:] for a in sys.argv if a.startswith("-D")]))
