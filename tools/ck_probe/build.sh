#!/bin/bash
# tools/ck_probe/build.sh -> tools/bin/libck_probe.so (gfx950).  ~60 CK kernels: one hipcc job per translation unit, 8 at a time.
set -e
cd "$(dirname "$0")"
python gen_bwd.py
mkdir -p _obj ../bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/opt/rocm/include -DCK_USE_XDL -Wno-unused-value"
# (a translation unit whose object is newer than every source of this directory is kept)
newest=$(ls -t *.hip *.h _gen/*.inc | head -1)
todo=$(for f in *.hip; do o=_obj/$(basename $f .hip).o; [ -f $o ] && [ $o -nt $newest ] || echo $f; done)
echo "compiling: $todo"
echo $todo | tr ' ' '\n' | grep . | xargs -P 8 -I{} sh -c "/opt/rocm/bin/hipcc $FLAGS -c {} -o _obj/\$(basename {} .hip).o 2> _obj/\$(basename {} .hip).log || (echo FAILED {}; tail -30 _obj/\$(basename {} .hip).log)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../bin/libck_probe.so _obj/*.o
ls -la ../bin/libck_probe.so
