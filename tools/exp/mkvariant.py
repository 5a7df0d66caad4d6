#!/usr/bin/env python3
"""Build tools/exp/libviterbi_<name>.so with extra -D flags: mkvariant.py <name> -DVIT_X=1 ..."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "viterbi.dll_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
print(b.build(force=True, extra=sys.argv[2:], out=os.path.join(ROOT, "tools", "exp", "libviterbi_%s.so" % sys.argv[1])))
