#!/usr/bin/env python3
"""glsl2cpp.py -- LEXICAL preprocessor that turns one of the reference's compute shaders into a C++ translation unit
for oracle/glsl/glsl_shim.h (TEST INFRASTRUCTURE; recipe side of oracle/_ref/, see oracle/glsl/README.md).

The shader text is read from the reference checkout where it lies and is never copied into this repository: the
generated .cpp goes to oracle/_ref/gen/ (git-ignored).  What this script does to the text -- and nothing else:

  1. `#include "x"` is replaced by the file's contents (looked up next to the including file, then in the shader root,
     the two places src/wgpu_utils/shader.rs:144-150 can resolve it from); `#version` / `#extension` lines are dropped;
     comments are blanked.
  2. Interface declarations `layout(...) uniform|buffer ...;` become C++ declarations + a registration record
     (GLSL_VOLUME / GLSL_BUFFER / uniform-block structs); `layout(local_size_x = ..) in;` becomes GLSL_LOCAL_SIZE.
  3. Parameter qualifiers: `inout T x` / `out T x` -> `T& x`, `in T x` -> `T x`.
  4. Floating-point literals are typed: `0.5` -> `F32(0.5f)` (GLSL literals are 32-bit floats; in C++ they would be
     doubles and promote the arithmetic).
  5. `float` is #defined to the shim's F32 and `shared` to `static` for the shader's text.

Function bodies, macros, control flow, operators and operand order are the reference's own characters.
"""
import os
import re
import sys

RES_TYPES = {"texture3D", "utexture3D", "image3D", "uimage3D", "texture2D"}
QUALIFIERS = {"restrict", "readonly", "writeonly", "coherent", "volatile"}


def strip_comments(text):
    out = []
    i, n = 0, len(text)
    while i < n:
        if text.startswith("//", i):
            j = text.find("\n", i)
            j = n if j < 0 else j
            # a line comment ending in a backslash would continue a macro: keep the continuation
            if j > i and text[j - 1] == "\\":
                out.append("\\")
            i = j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            j = n if j < 0 else j + 2
            out.append("".join(c if c == "\n" else " " for c in text[i:j]))
            i = j
        else:
            out.append(text[i])
            i += 1
    return "".join(out)


def flatten(path, root, stack=()):
    path = os.path.normpath(path)
    if path in stack:
        raise RuntimeError("recursive include of %s" % path)
    lines = []
    with open(path, encoding="utf-8") as f:
        text = f.read()
    for line in text.split("\n"):
        m = re.match(r'\s*#\s*include\s+"([^"]+)"', line)
        if m:
            cands = [os.path.join(os.path.dirname(path), m.group(1)), os.path.join(root, m.group(1))]
            for c in cands:
                if os.path.exists(c):
                    lines.append(flatten(c, root, stack + (path,)))
                    break
            else:
                raise RuntimeError("cannot resolve include %s from %s" % (m.group(1), path))
        elif re.match(r"\s*#\s*(version|extension)\b", line):
            lines.append("")
        else:
            lines.append(line)
    return "\n".join(lines)


def find_matching(text, i, open_c, close_c):
    depth = 0
    while i < len(text):
        if text[i] == open_c:
            depth += 1
        elif text[i] == close_c:
            depth -= 1
            if depth == 0:
                return i
        i += 1
    raise RuntimeError("unbalanced %s" % open_c)


def parse_members(body):
    """`T a; T b[]; T c[4];` -> [(type, name, array_suffix or None)]"""
    members = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        m = re.match(r"^(\w+)\s+(\w+)\s*(\[\s*\w*\s*\])?$", decl)
        if not m:
            raise RuntimeError("cannot parse block member %r" % decl)
        members.append((m.group(1), m.group(2), m.group(3)))
    return members


def convert_layouts(text):
    out = []
    i = 0
    pat = re.compile(r"\blayout\s*\(")
    while True:
        m = pat.search(text, i)
        if not m:
            out.append(text[i:])
            break
        out.append(text[i:m.start()])
        close = find_matching(text, m.end() - 1, "(", ")")
        quals = text[m.end():close]
        # the declaration runs to the first ';' outside braces
        j = close + 1
        depth = 0
        while True:
            c = text[j]
            if c == "{":
                depth += 1
            elif c == "}":
                depth -= 1
            elif c == ";" and depth == 0:
                break
            j += 1
        decl = text[close + 1:j].strip()
        newlines = "\n" * text[m.start():j + 1].count("\n") if "\\" not in text[m.start():j + 1] else ""
        out.append(convert_decl(quals, decl) + newlines)
        i = j + 1
    return "".join(out)


def convert_decl(quals, decl):
    if decl == "in":
        ls = dict((k.strip(), v.strip()) for k, v in (q.split("=") for q in quals.split(",")))
        return "GLSL_LOCAL_SIZE(%s, %s, %s)" % (ls.get("local_size_x", "1"), ls.get("local_size_y", "1"), ls.get("local_size_z", "1"))
    storage, rest = None, decl
    while True:   # storage keyword and memory qualifiers come in any order
        t = rest.split(None, 1)
        if t[0] in QUALIFIERS:
            rest = t[1]
        elif t[0] in ("uniform", "buffer") and storage is None:
            storage, rest = t[0], t[1]
        else:
            break
    if storage is None:
        raise RuntimeError("unsupported interface declaration: layout(%s) %s" % (quals, decl))
    if "{" not in rest:   # opaque resource:  TYPE NAME [n]
        m = re.match(r"^(\w+)\s+(\w+)\s*(?:\[\s*(\w+)\s*\])?$", rest.strip())
        if not m:
            raise RuntimeError("cannot parse resource %r" % rest)
        ty, name, arr = m.groups()
        if ty == "sampler":
            return "GLSL_SAMPLER(%s)" % name
        if ty not in RES_TYPES:
            raise RuntimeError("unknown resource type %s" % ty)
        if arr:
            return "GLSL_VOLUME_ARRAY(%s, %s, %s)" % (ty, name, arr)
        return "GLSL_VOLUME(%s, %s)" % (ty, name)
    m = re.match(r"^(\w+)\s*\{(.*)\}\s*(\w+)?$", rest.strip(), re.S)
    if not m:
        raise RuntimeError("cannot parse block %r" % rest)
    block, body, inst = m.groups()
    members = parse_members(body)
    if storage == "buffer":
        if len(members) == 1 and members[0][2] is not None and members[0][2].replace(" ", "") == "[]":
            return "GLSL_BUFFER(%s, %s)" % (members[0][0], members[0][1])
        if inst:
            raise RuntimeError("named buffer block instances are not supported (%s)" % block)
        s = "struct %s_t { %s }; GLSL_BLOCKPTR(%s_t, %s_p, %s) " % (block, " ".join("%s %s%s;" % (t, n, a or "") for t, n, a in members), block, block, block)
        # members of an instance-less block are globals: reach them through the bound pointer
        s += " ".join("\n#define %s (%s_p->%s)" % (n, block, n) for t, n, a in members) + "\n"
        return s
    # uniform / push_constant block
    struct = "struct %s_t { %s };" % (block, " ".join("%s %s%s;" % (t, n, a or "") for t, n, a in members))
    var = inst if inst else "%s_v" % block
    s = "%s static %s_t %s; " % (struct, block, var)
    for t, n, a in members:
        if inst:
            s += "static ::glsl::Reg _reg_%s_%s(_shader, \"%s.%s\", &%s.%s, sizeof(%s.%s), ::glsl::REG_UNIFORM); " % (var, n, inst, n, var, n, var, n)
        else:
            s += "GLSL_UNIFORM_ALIAS(%s, %s) GLSL_UNIFORM_MEMBER(%s, %s) " % (var, n, var, n)
    return s


FLOAT_LIT = re.compile(r"(?<![\w.])((?:\d+\.\d*|\.\d+)(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+)[fF]?(?![\w.])")


def convert(path, root, name):
    text = strip_comments(flatten(path, root))
    uses_barrier = bool(re.search(r"\bbarrier\s*\(", text))
    text = convert_layouts(text)
    text = re.sub(r"\b(?:inout|out)\s+(\w+)\s+(\w+)", r"\1& \2", text)
    text = re.sub(r"\bin\s+(\w+)\s+(\w+)", r"\1 \2", text)
    text = FLOAT_LIT.sub(lambda m: "F32(%sf)" % m.group(1), text)
    head = (
        "// GENERATED by oracle/glsl/glsl2cpp.py from %s -- do not commit (contains the reference's shader text)\n"
        "#include \"glsl_shim.h\"\n"
        "namespace ref_%s {\n"
        "using namespace glsl;\n"
        "static ::glsl::Shader _shader(\"%s\");\n"
        "#define COMPUTE_SHADER 1\n#define VERTEX_SHADER 0\n#define FRAGMENT_SHADER 0\n#define NDEBUG 1\n"
        "#define float F32\n#define shared static\n"
    ) % (os.path.relpath(path, root), name, name)
    tail = (
        "\n#undef float\n#undef shared\n"
        "static ::glsl::Entry _entry(_shader, &main, gl_WorkGroupSize, %s);\n"
        "}  // namespace\n"
    ) % ("true" if uses_barrier else "false")
    return head + text + tail


def main(argv):
    if len(argv) != 4:
        print("usage: glsl2cpp.py <shader root> <shader path relative to root> <out.cpp>", file=sys.stderr)
        return 2
    root, rel, out = argv[1:]
    name = os.path.splitext(os.path.basename(rel))[0]
    cpp = convert(os.path.join(root, rel), root, name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w", encoding="utf-8") as f:
        f.write(cpp)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
