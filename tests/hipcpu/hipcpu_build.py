"""TEST INFRASTRUCTURE -- compile a csrc kernel file for the HOST (g++, one OS thread per lane; see hip/hip_runtime.h)
so its logic can be compared with the oracle where there is no GPU.  Never imported by the package."""
import ctypes
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "transferattack_amd", "csrc")
OUT = os.path.join(HERE, "_build")

# name and element type of each file's dynamic-LDS array (a block-scope `extern` needs a namespace-scope definition)
_DYNAMIC_LDS = {"dim.hip": [("char", "smem_raw")], "tim.hip": [("float", "smem")], "bsr.hip": [("int", "plans")]}


def _host_text(text):
    text = text.replace("extern __shared__", "extern" if SANITIZE else "extern thread_local")
    text = text.replace('#include "../../include/ta_hip.h"', '#include "%s"' % os.path.join(ROOT, "include", "ta_hip.h"))
    # clang's ext_vector_type has no g++ counterpart with .x/.y members
    text = text.replace("typedef float floatx4 __attribute__((ext_vector_type(4)));", "struct floatx4 { float x, y, z, w; };")
    text = text.replace("typedef float v2f __attribute__((ext_vector_type(2)));", "struct alignas(8) v2f { float x, y; };")
    text = text.replace("typedef float f32x4 __attribute__((ext_vector_type(4)));", "/* f32x4: tests/hipcpu/hip/hip_runtime.h */")
    text = text.replace("typedef float stem_f32x4 __attribute__((ext_vector_type(4)));", "typedef f32x4 stem_f32x4;")
    # the MFMA instruction's arithmetic on the host (the product source only knows hipcc's device and host passes)
    text = text.replace("return c;                                         // hipcc's host pass only parses this function",
                        "return hipcpu_mfma_f32_16x16x4(a, b, c);")
    return text


HOST_SOURCES = ("runtime.hip", "update.hip", "elementwise.hip", "tim.hip", "dim.hip", "sia.hip", "bsr.hip", "spectrum.hip", "glue.hip", "stem.hip")


# HIPCPU_SANITIZE=1: AddressSanitizer build (one worker thread, `__shared__` arrays as plain statics).  Every load /
# store of a kernel to a tensor (torch / numpy heap blocks carry ASan red zones) is checked to the byte -- finer than the
# guard-page tests.  The LDS arrays of TEMPLATE kernels are COMDAT statics, which ASan does not instrument, so an LDS
# index out of range is still only caught through wrong results.  Run as
#   LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
#   HIPCPU_SANITIZE=1 python -m pytest tests/test_kernel_logic_host.py
SANITIZE = os.environ.get("HIPCPU_SANITIZE", "0") == "1"


def build():
    """-> path of libta_host.so: the kernel sources of csrc compiled for the host; rebuilt when a source is newer."""
    os.makedirs(OUT, exist_ok=True)
    lib = os.path.join(OUT, "libta_host_asan.so" if SANITIZE else "libta_host.so")
    headers = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")]
    sources = [os.path.join(CSRC, s) for s in HOST_SOURCES]
    deps = sources + headers + [os.path.join(HERE, "hipcpu.cpp"), os.path.join(HERE, "hip", "hip_runtime.h"), __file__]
    stale = lambda: not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps)   # noqa: E731
    if not stale():
        return lib
    import fcntl
    with open(os.path.join(OUT, ".build.lock"), "w") as lock:      # pytest-xdist workers: one builds, the others wait and find it done
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not stale():
            return lib
        for h in headers:
            with open(os.path.join(OUT, os.path.basename(h)), "w") as fh:
                fh.write(_host_text(open(h).read()))
        generated = []
        for src in sources:
            name = os.path.basename(src)
            text = open(src).read()
            # the dynamic-LDS array becomes a namespace-scope array defined BEFORE the kernels, and the kernels' block-scope
            # `extern __shared__` declarations of it are dropped: g++ reaches a block-scope `extern thread_local` through
            # its TLS init function -- a weak symbol that does not exist for a plain array -- and inside loops it omits
            # the null check (a call through address 0)
            for ctype, var in _DYNAMIC_LDS.get(name, []):
                text, found = re.subn(r"^[ \t]*extern __shared__[^;\n]*\b%s\[\];[^\n]*$" % var, "", text, flags=re.M)
                assert found, (name, var)
                text = "namespace ta { %s __attribute__((aligned(16))) %s %s[163840 / sizeof(%s)]; }\n" % (
                    "" if SANITIZE else "thread_local", ctype, var, ctype) + text
            text = _host_text(text)
            generated.append(os.path.join(OUT, os.path.splitext(name)[0] + "_host.cpp"))
            with open(generated[-1], "w") as fh:
                fh.write(text)
        extra = ["-fsanitize=address", "-fno-omit-frame-pointer", "-g", "-DHIPCPU_SINGLE_WORKER"] if SANITIZE else []
        cmd = ["g++", "-O1" if SANITIZE else "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
               "-pthread", "-Wno-attributes", "-Wno-unknown-pragmas", "-I", HERE, "-I", OUT] + extra + generated + [
               os.path.join(HERE, "hipcpu.cpp"), "-o", lib + ".%d.tmp" % os.getpid()]
        subprocess.run(cmd, check=True, capture_output=True, text=True)
        os.replace(cmd[-1], lib)          # a new inode: another pytest-xdist worker that has the old file mapped keeps its pages
    return lib


def load(tag=None):
    """ctypes.CDLL of the host build; ``tag`` loads a private copy (own statics, e.g. a variant knob read once)."""
    lib = build()
    if tag is not None:
        import shutil
        private = lib.replace(".so", "_%s.so" % tag)
        shutil.copyfile(lib, private + ".%d.tmp" % os.getpid())
        os.replace(private + ".%d.tmp" % os.getpid(), private)     # never write INTO a file another worker may have loaded
        lib = private
    return ctypes.CDLL(lib)
