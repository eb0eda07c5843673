#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- build oracle/_ref/: the reference's own host code, compiled from the
sources where they lie under /root/reference (nothing is copied into this repository).

    python oracle/ref_recipe/build_ref.py [--reference /root/reference]

Outputs (git-ignored, but they travel to the GPU box with the snapshot):
    oracle/_ref/libezrt_ref_p2.so   chapter 2 main.cpp  (hitTriangle/hitAABB/hitTriangleArray/hitBVH,
                                    pointer-tree builders, main()'s probe ray)
    oracle/_ref/libezrt_ref_p3.so   chapter 3 main.cpp + lib/hdrloader.cpp (readObj, getTransformMatrix,
                                    flat buildBVH/buildBVHwithSAH, main() headless, display() camera)
    oracle/_ref/libezrt_ref_p4.so   chapter 4 (Material defaults of P4/P5)
    oracle/_ref/libezrt_ref_p5.so   chapter 5 (+ calculateHdrCache)
    oracle/_ref/libezrt_ref_fsh_p{3,4,5}.so   the chapters' FRAGMENT SHADERS (shaders/fshader.fsh) compiled by g++: the
                                    shader text goes through the syntax-only pass of fsh_pass.py into a temporary
                                    file that wrap_fsh.cpp #includes against shim/glsl_shim.h (the GLSL language)
The reference's build system (CMake + GLM/GLEW/freeglut) is NOT used: each chapter is one
main.cpp, compiled by g++ directly against the stand-in headers in oracle/ref_recipe/shim/.
Same arithmetic contract as the rest of the repo: -O2 -ffp-contract=off -fno-fast-math.
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "_ref")

CHAPTERS = {
    "p2": ("part 2 -- BVH Accelerate Struct", "wrap_p2.cpp", False),
    "p3": ("part 3 -- OpenGL Raytracing", "wrap_flat.cpp", False),
    "p4": ("part 4 -- Disney Principle BRDF", "wrap_flat.cpp", False),
    "p5": ("part 5 -- Importance Sampling & Low Discrepancy Sequence", "wrap_flat.cpp", True),
}


def source_dir(ref, tag):
    return os.path.join(ref, CHAPTERS[tag][0], "source code")


def build(ref="/root/reference", verbose=True):
    if not os.path.isdir(ref):
        raise FileNotFoundError(ref)
    os.makedirs(OUT, exist_ok=True)
    built = []
    for tag, (_, wrap, has_cache) in CHAPTERS.items():
        src = source_dir(ref, tag)
        main_cpp = os.path.join(src, "main.cpp")
        out = os.path.join(OUT, "libezrt_ref_%s.so" % tag)
        wrap_path = os.path.join(HERE, wrap)
        deps = [main_cpp, wrap_path, os.path.join(HERE, "shim", "glm", "glm.hpp"),
                os.path.join(HERE, "shim", "GL", "glew.h"), os.path.abspath(__file__)]
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-fno-fast-math",
               "-w", "-I", os.path.join(HERE, "shim"), '-DEZRT_REF_MAIN="%s"' % main_cpp]
        if wrap == "wrap_flat.cpp":
            hdr = os.path.join(src, "lib", "hdrloader.cpp")
            deps.append(hdr)
            cmd.append('-DEZRT_REF_HDRLOAD="%s"' % hdr)
        if has_cache:
            cmd.append("-DEZRT_REF_HAS_HDRCACHE=1")
        cmd += ["-o", out, wrap_path]
        if wrap == "wrap_flat.cpp":
            cmd.append(os.path.join(HERE, "wrap_hdr.cpp"))
            deps.append(os.path.join(HERE, "wrap_hdr.cpp"))
        if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
            built.append(out)
            continue
        if verbose:
            print("[oracle/_ref] g++ %s <- %s" % (os.path.basename(out), main_cpp))
        subprocess.check_call(cmd)
        built.append(out)
    built += build_fsh(ref, verbose)
    return built


def build_fsh(ref, verbose=True):
    """libezrt_ref_fsh_p{3,4,5}.so: each chapter's fragment shader, compiled (see wrap_fsh.cpp / fsh_pass.py)."""
    sys.path.insert(0, HERE)
    import fsh_pass
    repo = os.path.dirname(os.path.dirname(HERE))
    built = []
    for ch in (3, 4, 5):
        fsh = os.path.join(source_dir(ref, "p%d" % ch), "shaders", "fshader.fsh")
        out = os.path.join(OUT, "libezrt_ref_fsh_p%d.so" % ch)
        deps = [fsh, os.path.join(HERE, "wrap_fsh.cpp"), os.path.join(HERE, "fsh_pass.py"), os.path.join(HERE, "shim", "glsl_shim.h"),
                os.path.join(repo, "include", "ezrt_detmath.h"), os.path.abspath(__file__)]
        if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
            built.append(out)
            continue
        tmp = tempfile.mkdtemp(prefix="ezrt_fsh_")
        try:
            inc = os.path.join(tmp, "fsh_p%d.inc" % ch)
            with open(inc, "w", encoding="utf-8") as f:
                f.write(fsh_pass.translate(open(fsh, encoding="utf-8").read(), ch))
            cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fno-math-errno",
                   "-fsingle-precision-constant", "-w", "-I", os.path.join(HERE, "shim"), "-I", os.path.join(repo, "include"),
                   '-DEZRT_REF_FSH="%s"' % inc, "-DEZRT_FSH_CHAPTER=%d" % ch, "-o", out, os.path.join(HERE, "wrap_fsh.cpp")]
            if verbose:
                print("[oracle/_ref] g++ %s <- %s" % (os.path.basename(out), fsh))
            subprocess.check_call(cmd)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)   # the translated shader text is not kept anywhere
        built.append(out)
    return built


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    a = ap.parse_args()
    try:
        for p in build(a.reference):
            print(p)
    except FileNotFoundError:
        print("reference not present at %s: keeping prebuilt oracle/_ref as it is" % a.reference)
        sys.exit(0)
