"""Per-kernel register / LDS / scratch budget of libgsfm's HIP sources, read off the compiler's own metadata.

Compiles each `glomap_amd/csrc/*.hip` for gfx950 to assembly (device side only, the flags of glomap_amd/build.py) and
prints VGPR / AGPR / SGPR counts, static LDS, scratch and spills per kernel, plus the resident waves per SIMD the VGPR
count allows (512 VGPRs per SIMD lane, allocation granule 8, at most 8 waves).  With `--asm DIR` the .s files are kept
for reading (this is how the dependent-load chain of k_gp_phaseA was found).  No GPU needed.
Usage: python tools/kernel_resources.py [--asm DIR] [file.hip ...]"""
import pathlib
import re
import subprocess
import sys
import tempfile

ROOT = pathlib.Path(__file__).resolve().parent.parent
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "--offload-device-only", "-S"]


def waves_per_simd(vgpr, agpr):
    total = max(1, -(-(int(vgpr) + int(agpr)) // 8) * 8)  # unified register file on CDNA3/4
    return max(1, min(8, 512 // total))


def main():
    args = sys.argv[1:]
    keep = None
    if args[:1] == ["--asm"]:
        keep = pathlib.Path(args[1])
        keep.mkdir(parents=True, exist_ok=True)
        args = args[2:]
    srcs = [pathlib.Path(a) for a in args] or sorted((ROOT / "glomap_amd" / "csrc").glob("*.hip"))
    with tempfile.TemporaryDirectory() as tmp:
        out_dir = keep or pathlib.Path(tmp)
        for src in srcs:
            asm = out_dir / (src.stem + ".s")
            subprocess.run(["hipcc", *FLAGS, str(src), f"-I{ROOT / 'include'}", "-o", str(asm)], check=True)
            text = asm.read_text()
            print(f"== {src.name}")
            for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size: +\d+", text, re.S):
                blk = m.group(0)
                name = re.search(r"\.name: +(\S+)", blk).group(1)
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                dem = dem.replace("(anonymous namespace)::", "").replace("gsfm::", "").replace("void ", "")
                dem = re.sub(r"\(.*", "", dem)

                def g(key):
                    mm = re.search(r"\." + key + r": +(\d+)", blk)
                    return int(mm.group(1)) if mm else 0

                v, a = g("vgpr_count"), g("agpr_count")
                print(f"  {dem:40s} vgpr {v:4d} agpr {a:4d} sgpr {g('sgpr_count'):4d} lds {g('group_segment_fixed_size'):6d} "
                      f"scratch {g('private_segment_fixed_size'):5d} spills {g('vgpr_spill_count'):3d}  waves/SIMD {waves_per_simd(v, a)}")


if __name__ == "__main__":
    main()
