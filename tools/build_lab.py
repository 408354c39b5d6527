"""Lab builds of the library: tools/_bin/libwxhip_<name>.so compiled with extra -D flags (ablations: LAB_NO_W -- the
weight loads of the decode GEMVs hit one cached tile; LAB_NO_SELFKV -- the decoder self-attention reads one cached key;
DL_POLL_OVERRIDE=n, DL_NO_FALLBACK -- declayer.hip).  Results of such builds are wrong on purpose; they answer "what
does this traffic cost".   python tools/build_lab.py name FLAG [FLAG ...]   then   python tools/ab_lib.py K - tools/_bin/libwxhip_name.so"""
import concurrent.futures, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperx_mlx_amd import build as B

name, flags = sys.argv[1], ["-D" + f for f in sys.argv[2:]]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin")
obj = os.path.join(out, "_obj_" + name)
os.makedirs(obj, exist_ok=True)

def cc(src):
    o = os.path.join(obj, os.path.basename(src)[:-4] + ".o")
    subprocess.run([B._hipcc(), *B.FLAGS, *B.EXTRA_FLAGS.get(os.path.basename(src), []), *flags, "-c", src, "-o", o], check=True)
    return o

with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
    objs = list(ex.map(cc, B.sources()))
lib = os.path.join(out, f"libwxhip_{name}.so")
subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", *objs, "-o", lib], check=True)
print(lib)
