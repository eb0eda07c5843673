#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- the mechanical source pass that lets g++ compile the reference's fragment shaders.

translate(text, chapter) takes the text of `part {3,4,5} .../source code/shaders/fshader.fsh` as read from
/root/reference and returns C++ that wrap_fsh.cpp #includes INSIDE `struct Fsh : GlslBuiltins { ... };` (shim/glsl_shim.h
supplies the language).  The result is written to a temporary directory by build_ref.py and deleted after the compile:
nothing of the reference is kept in this repository or in oracle/_ref/ except the compiled object code.

What the pass does -- syntax only, every rule listed here, each asserted to apply where expected:
  1. comments and the #version line are removed;
  2. storage qualifiers: `uniform T x;` / `in T x;` -> `inline static T x;` (set by the wrapper before a sample),
     `out T x;` -> `T x;`; parameter qualifiers `inout T x` -> `T& x`, `in T x` -> `T x`;
  3. swizzles with more than one component become calls: `.xyz` -> `.xyz()`, likewise rgb, rg, xy (read-only uses);
  4. GLSL evaluates function-call arguments left to right, C++ leaves the order unspecified: a statement with more
     than one `rand()` gets its draws hoisted into named temporaries in textual order
     (P5/fsh:822 `SampleHdr(rand(), rand())`, :923 the AA jitter; the same AA line in P3/P4);
  5. the frame's maxBounce literal in main() becomes the wrapper's variable `ezrt_max_bounce` (P3: 2, P4: 4, P5: 2 are
     the reference's values; the configs of BASELINE.json use other counts), and for chapter 5 the wrapper's flag
     `ezrt_use_is` selects between the two integrator calls the reference switches by (un)commenting
     (P5/fsh:936-937: `pathTracingImportanceSampling` active, `pathTracing` commented out);
  6. `gl_FragData[0]` is declared by the wrapper; file-scope initialisers (`uint seed = ...`) become default member
     initialisers of the struct, i.e. they are re-evaluated for every pixel-sample, as a fragment invocation does.
A canary at the end: no statement with two `rand()` is left, no GLSL keyword is left, every patch of rule 5 matched
exactly once.
"""
import re


class PassError(RuntimeError):
    pass


def _strip_comments(s):
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    return re.sub(r"//[^\n]*", "", s)


def _sub_once(pattern, repl, s, what):
    s2, n = re.subn(pattern, repl, s)
    if n != 1:
        raise PassError("fsh_pass: %s matched %d times (expected exactly once) -- the reference text changed" % (what, n))
    return s2


def translate(text, chapter):
    raw = text
    s = _strip_comments(text)
    s = re.sub(r"^\s*#version[^\n]*\n", "", s, flags=re.M)

    # ---- rule 5 first (it reads the comments): chapter 5's two integrator calls
    if chapter == 5:
        if not re.search(r"vec3 Li = pathTracingImportanceSampling\(firstHit, maxBounce\);\s*\n\s*//vec3 Li = pathTracing\(firstHit, maxBounce\);", raw):
            raise PassError("fsh_pass: P5 main() no longer holds the active IS call followed by the commented uniform call")
        s = _sub_once(r"int maxBounce = 2;", "int maxBounce = ezrt_max_bounce;", s, "P5 `int maxBounce = 2;`")
        s = _sub_once(r"vec3 Li = pathTracingImportanceSampling\(firstHit, maxBounce\);",
                      "vec3 Li = ezrt_use_is ? pathTracingImportanceSampling(firstHit, maxBounce) : pathTracing(firstHit, maxBounce);",
                      s, "P5 integrator call")
    elif chapter == 4:
        s = _sub_once(r"pathTracing\(firstHit, 4\)", "pathTracing(firstHit, ezrt_max_bounce)", s, "P4 `pathTracing(firstHit, 4)`")
    elif chapter == 3:
        s = _sub_once(r"pathTracing\(firstHit, 2\)", "pathTracing(firstHit, ezrt_max_bounce)", s, "P3 `pathTracing(firstHit, 2)`")
    else:
        raise PassError("chapter must be 3, 4 or 5")

    # ---- rule 2: qualifiers
    s, n_uni = re.subn(r"^\s*uniform\s+(\w+)\s+(\w+)\s*;", r"inline static \1 \2;", s, flags=re.M)
    s, n_in = re.subn(r"^\s*in\s+(\w+)\s+(\w+)\s*;", r"inline static \1 \2;", s, flags=re.M)
    s, n_out = re.subn(r"^\s*out\s+(\w+)\s+(\w+)\s*;", r"\1 \2;", s, flags=re.M)
    if n_in != 1 or n_out != 1 or n_uni < 10:
        raise PassError("fsh_pass: expected `in vec3 pix;`, `out vec4 fragColor;` and the uniform block (%d/%d/%d)" % (n_in, n_out, n_uni))
    s = re.sub(r"\binout\s+(\w+)\s+(\w+)", r"\1& \2", s)
    s = re.sub(r"([(,]\s*)in\s+(\w+)\s+(\w+)", r"\1\2 \3", s)

    # ---- rule 3: multi-component swizzles (all uses are reads)
    s = re.sub(r"\.(xyz|rgb|rg|xy)\b(?!\s*\()", r".\1()", s)
    if re.search(r"\.(xyz|rgb|rg|xy)\(\)\s*[-+*/]?=[^=]", s):
        raise PassError("fsh_pass: a swizzle is assigned to; the shim only implements reads")

    # ---- rule 4: sequence the rand() draws of a statement left to right
    out, k = [], 0
    for line in s.split("\n"):
        n = line.count("rand()")
        if n > 1:
            if line.count(";") != 1 or "for" in line:
                raise PassError("fsh_pass: cannot hoist the rand() draws of: " + line.strip())
            indent = re.match(r"\s*", line).group(0)
            names = []
            for _ in range(n):
                names.append("ezrt_draw%d" % k)
                out.append("%sfloat ezrt_draw%d = rand();" % (indent, k))
                k += 1
            for name in names:
                line = line.replace("rand()", name, 1)
        out.append(line)
    s = "\n".join(out)
    if k not in (2, 4):
        raise PassError("fsh_pass: hoisted %d draws (expected 2 in P3/P4, 4 in P5)" % k)

    # ---- canary
    for line in s.split("\n"):
        if line.count("rand()") > 1:
            raise PassError("fsh_pass: a statement with two rand() calls survived")
    for kw in ("uniform", "inout", "#version"):
        if re.search(r"(^|\W)%s\W" % re.escape(kw), s):
            raise PassError("fsh_pass: GLSL keyword '%s' survived the pass" % kw)
    if re.search(r"[(,]\s*(in|out)\s+\w+\s+\w+", s):
        raise PassError("fsh_pass: a parameter qualifier survived the pass")
    return s


if __name__ == "__main__":
    import sys
    print(translate(open(sys.argv[1], encoding="utf-8").read(), int(sys.argv[2])))
