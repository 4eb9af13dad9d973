"""Register / LDS / scratch usage of the kernels inside libsmelter_hip.so (or any HIP fat binary): extracts the gfx950 code objects from the
.hip_fatbin section (clang-offload-bundler format) and prints the AMDGPU metadata notes.
    python tools/kernel_resources.py [--lib smelter_amd/libsmelter_hip.so] [substring ...]"""
import os, re, subprocess, sys, tempfile
lib = "smelter_amd/libsmelter_hip.so"
args = sys.argv[1:]
if args and args[0] == "--lib":
    lib = args[1]; args = args[2:]
LLVM = "/opt/rocm/lib/llvm/bin/"
data = open(lib, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
pos, n = 0, 0
with tempfile.TemporaryDirectory() as td:
    while True:
        pos = data.find(magic, pos)
        if pos < 0:
            break
        import struct
        nb = struct.unpack_from("<Q", data, pos + 24)[0]
        off = pos + 32
        for _ in range(nb):
            o, sz, tl = struct.unpack_from("<QQQ", data, off); off += 24
            triple = data[off:off + tl].decode(); off += tl
            if "gfx950" in triple and sz:
                path = os.path.join(td, "co%d.o" % n); n += 1
                open(path, "wb").write(data[pos + o:pos + o + sz])
                txt = subprocess.run([LLVM + "llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
                for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
                    nm = re.search(r"\.name:\s+(\S+)", blk)
                    if not nm:
                        continue
                    dem = subprocess.run(["c++filt", nm.group(1)], capture_output=True, text=True).stdout.strip()
                    if args and not any(a in dem for a in args):
                        continue
                    g = lambda k: int(re.search(k + r":\s+(\d+)", blk).group(1))
                    print("%-110s agpr %3d vgpr %3d sgpr %3d spill %3d scratch %5d lds %6d" % (dem[:110], int(blk.split()[0]), g(r"\.vgpr_count"), g(r"\.sgpr_count"),
                          g(r"\.vgpr_spill_count"), g(r"\.private_segment_fixed_size"), g(r"\.group_segment_fixed_size")))
        pos += 24
