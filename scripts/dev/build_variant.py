"""Development: build a variant of the library that differs in ONE unit only (extra -D defines; et_forward_tile.hip, or the unit
named by VARIANT_UNIT), linked against the regular objects of the other units.
    [VARIANT_UNIT=et_backward_tile.hip] python scripts/dev/build_variant.py NAME -DFOO [-DBAR ...]
-> epipolar_transformers_amd/lib/libepipolar_amd_NAME.so  (load it through EPIPOLAR_AMD_LIB)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from epipolar_transformers_amd import build as b

name, defs = sys.argv[1], sys.argv[2:]
unit = os.environ.get("VARIANT_UNIT", "et_forward_tile.hip")
obj = os.path.join(b.OBJ + "_" + name, unit.replace(".hip", ".o"))
os.makedirs(os.path.dirname(obj), exist_ok=True)
cmd = [b.hipcc()] + b.flags() + defs + ["-c", "-o", obj, b._path(unit)]
subprocess.check_call(cmd)
lib = os.path.join(b.PKG, "lib", "libepipolar_amd_%s.so" % name)
objs = [obj if u == unit else b._obj(u) for u in b.UNITS]
subprocess.check_call([b.hipcc(), "--offload-arch=" + b.ARCH, "-shared", "-fPIC", "-o", lib] + objs)
print(lib)
