"""CPU test: the Java side of the boundary (java/org/apache/pinot/gpu/*.java) agrees with the C side (include/pinot_gpu.h,
jni/pg_marshal.h) on every number that crosses it.

No JDK exists here, so the Java sources are never compiled or run; a swapped enum value there (round 2: PRED_IS_NULL / PRED_DOC_RANGE)
would silently turn every sorted-column predicate into IS NULL in a real server.  This test parses both languages:

  * every `static final int NAME = <literal>` of PinotGpuNative.java whose NAME starts with PG_ / PGM_ must exist under the same name in
    the C headers with the same value, and every enumerator / #define of the C headers that the Java side can send or receive must be
    mirrored there;
  * no OTHER Java class may define a boundary constant with a numeric literal: they have to reference PinotGpuNative.<NAME>, and the
    alias must point at the constant its name says (PRED_X -> PG_PRED_X, OP_X -> PG_FILTER_X, ...);
  * record sizes and result slots are never spelled as bare numbers next to the flat arrays (`3 * n`, `Object[8]`, `result[6]`);
  * the JNI function names / signatures of jni/pinot_gpu_jni.c match the `static native` declarations.
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JAVA_DIR = os.path.join(ROOT, "java", "org", "apache", "pinot", "gpu")

# Java alias prefix -> C prefix (an alias may also use the full C name)
ALIAS_PREFIXES = [("PRED_", "PG_PRED_"), ("EVAL_", "PG_EVAL_"), ("OP_", "PG_FILTER_"), ("AGG_", "PG_AGG_"), ("TYPE_", "PG_TYPE_"),
                  ("FWD_", "PG_FWD_"), ("H_", "PGM_H_"), ("R_", "PGM_R_")]
# C names the Java side must mirror (prefix families); PG_KERNEL_* / PG_CFG_* are diagnostics Java never reads
MIRRORED_FAMILIES = ("PG_OK", "PG_ERR_", "PG_TYPE_", "PG_FWD_", "PG_PRED_", "PG_EVAL_", "PG_FILTER_", "PG_AGG_", "PG_QUERY_NULL_HANDLING", "PG_QUERY_STATS_UPPER_BOUND_OK",
                     "PG_ABI_VERSION", "PGM_")


def strip_c_comments(text):
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S))


def c_constants(text=None):
    """NAME -> int from the enums and integer #defines of the two headers."""
    out = {}
    texts = [text] if text is not None else [open(os.path.join(ROOT, p)).read() for p in ("include/pinot_gpu.h", "jni/pg_marshal.h")]
    for t in texts:
        t = strip_c_comments(t)
        for name, value in re.findall(r"#define\s+(PGM?_[A-Z0-9_]+)\s+(-?\d+)\b", t):
            out[name] = int(value)
        for body in re.findall(r"enum\s*\w*\s*\{(.*?)\}", t, flags=re.S):
            for name, value in re.findall(r"\b(PGM?_[A-Z0-9_]+)\s*=\s*(-?\d+)", body):
                out[name] = int(value)
    return out


def java_files(directory=JAVA_DIR):
    return {f: open(os.path.join(directory, f)).read() for f in sorted(os.listdir(directory)) if f.endswith(".java")}


def java_int_constants(text):
    """(NAME, right-hand side) of every `static final int NAME = ...;`"""
    return re.findall(r"static\s+final\s+int\s+([A-Z][A-Z0-9_]*)\s*=\s*([^;]+);", strip_c_comments(text))


def check_java_against_c(files, c):
    """Returns the list of disagreements (empty = the boundary agrees with itself)."""
    problems = []
    native = files.get("PinotGpuNative.java", "")
    mirrored = {}
    for name, rhs in java_int_constants(native):
        if name.startswith(("PG_", "PGM_")):
            if not re.fullmatch(r"-?\d+", rhs.strip()):
                problems.append("PinotGpuNative.%s is not a literal" % name)
                continue
            mirrored[name] = int(rhs)
            if name not in c:
                problems.append("PinotGpuNative.%s does not exist in the C headers" % name)
            elif c[name] != int(rhs):
                problems.append("PinotGpuNative.%s = %s but the C headers say %d" % (name, rhs.strip(), c[name]))
    for name, value in c.items():
        if name.startswith(MIRRORED_FAMILIES) and name not in mirrored:
            problems.append("%s = %d of the C headers has no mirror in PinotGpuNative.java" % (name, value))
    for fname, text in files.items():
        for name, rhs in java_int_constants(text):
            rhs = rhs.strip()
            target = None
            for jp, cp in ALIAS_PREFIXES:
                if name.startswith(jp) and (cp + name[len(jp):]) in c:
                    target = cp + name[len(jp):]
            if fname != "PinotGpuNative.java" and name in c:
                target = name
            if target is None:
                continue
            if re.fullmatch(r"-?\d+", rhs):
                if fname != "PinotGpuNative.java":
                    problems.append("%s: %s = %s is a boundary constant spelled as a literal (C: %s = %d)" % (fname, name, rhs, target, c[target]))
            elif rhs != "PinotGpuNative." + target:
                problems.append("%s: %s = %s should be PinotGpuNative.%s" % (fname, name, rhs, target))
    return problems


def test_java_constants_equal_the_c_headers():
    c = c_constants()
    assert c["PG_PRED_DOC_RANGE"] == 5 and c["PG_PRED_IS_NULL"] == 6 and c["PGM_HEADER_LEN"] == 13      # the parser sees the headers
    problems = check_java_against_c(java_files(), c)
    assert not problems, "\n".join(problems)


def test_the_check_catches_the_round_2_swap():
    """The defect this test exists for, replayed: PRED_IS_NULL = 5 / PRED_DOC_RANGE = 6 as literals in GpuQueryLowering."""
    c = c_constants()
    files = java_files()
    broken = dict(files)
    broken["GpuQueryLowering.java"] = files["GpuQueryLowering.java"].replace(
        "PRED_DOC_RANGE = PinotGpuNative.PG_PRED_DOC_RANGE", "PRED_DOC_RANGE = 6").replace(
        "PRED_IS_NULL = PinotGpuNative.PG_PRED_IS_NULL", "PRED_IS_NULL = 5")
    assert broken["GpuQueryLowering.java"] != files["GpuQueryLowering.java"]
    problems = check_java_against_c(broken, c)
    assert any("PRED_IS_NULL" in p for p in problems) and any("PRED_DOC_RANGE" in p for p in problems)
    swapped = dict(files)
    swapped["PinotGpuNative.java"] = files["PinotGpuNative.java"].replace("PG_PRED_DOC_RANGE = 5", "PG_PRED_DOC_RANGE = 6").replace(
        "PG_PRED_IS_NULL = 6", "PG_PRED_IS_NULL = 5")
    problems = check_java_against_c(swapped, c)
    assert any("PG_PRED_IS_NULL" in p for p in problems)
    crossed = dict(files)
    crossed["GpuQueryLowering.java"] = files["GpuQueryLowering.java"].replace(
        "PRED_IS_NULL = PinotGpuNative.PG_PRED_IS_NULL", "PRED_IS_NULL = PinotGpuNative.PG_PRED_DOC_RANGE")
    assert any("PRED_IS_NULL" in p for p in check_java_against_c(crossed, c))


def test_flat_array_records_are_never_bare_numbers():
    """`3 * nodes`, `new Object[8]`, `result[6]`: record sizes and result slots must be the named constants on both sides."""
    for fname, text in java_files().items():
        if fname == "PinotGpuNative.java":
            continue
        code = strip_c_comments(text)
        assert not re.search(r"\b(result|raw)\s*\[\s*\d+\s*\]", code), fname + ": result slot spelled as a number"
        assert not re.search(r"new\s+Object\s*\[\s*\d+\s*\]", code), fname
        for m in re.finditer(r"\b(\d+)\s*\*\s*(\w+)", code):
            if m.group(2) in ("numPredicates", "i", "n", "p", "a", "c") and int(m.group(1)) in (2, 3, 4, 6, 8):
                # the only arithmetic of this shape left in the package is bit twiddling (e.g. `4 * d` of a byte offset)
                line = code[code.rfind("\n", 0, m.start()) + 1:code.find("\n", m.end())]
                assert "putInt" in line or "allocateDirect" in line, fname + ": " + line.strip()
    jni = strip_c_comments(open(os.path.join(ROOT, "jni", "pinot_gpu_jni.c")).read())
    assert not re.search(r"NewObjectArray\(env,\s*\d+", jni)
    assert not re.search(r"SetObjectArrayElement\(env,\s*out,\s*\d+", jni)
    assert not re.search(r"GetArrayLength\([^)]*\)\s*/\s*\d+", jni) and not re.search(r"!=\s*\d+\s*\*\s*num_", jni)
    marshal = strip_c_comments(open(os.path.join(ROOT, "jni", "pg_marshal.c")).read())
    assert not re.search(r"\[\s*\d+\s*\*\s*[a-z]\b", marshal), "pg_marshal.c indexes a flat array with a bare record size"


JNI_TYPES = {"void": "void", "int": "jint", "long": "jlong", "String": "jstring", "ByteBuffer": "jobject", "int[]": "jintArray",
             "long[]": "jlongArray", "String[]": "jobjectArray", "Object[]": "jobjectArray", "Object[][]": "jobjectArray"}


def test_jni_functions_match_the_native_declarations():
    java = strip_c_comments(java_files()["PinotGpuNative.java"])
    jni = strip_c_comments(open(os.path.join(ROOT, "jni", "pinot_gpu_jni.c")).read())
    natives = re.findall(r"static\s+native\s+([\w\[\]]+)\s+(\w+)\s*\(([^)]*)\)\s*;", java)
    assert len(natives) >= 10
    for ret, name, params in natives:
        m = re.search(r"JNIEXPORT\s+(\w+)\s+JNICALL\s+Java_org_apache_pinot_gpu_PinotGpuNative_%s\s*\(([^)]*)\)" % name, jni)
        assert m, "no JNI function for PinotGpuNative." + name
        assert m.group(1) == JNI_TYPES[ret], name
        want = ["JNIEnv*", "jclass"] + [JNI_TYPES[p.strip().rsplit(" ", 1)[0].strip()] for p in params.split(",") if p.strip()]
        got = [re.sub(r"\s*\w+$", "", p.strip()) for p in m.group(2).split(",")]
        assert got == want, (name, got, want)
    exported = set(re.findall(r"Java_org_apache_pinot_gpu_PinotGpuNative_(\w+)\s*\(", jni))
    assert exported == {n for _, n, _ in natives}


def _strip_java_literals(text):
    text = strip_c_comments(text)
    text = re.sub(r'"(\\.|[^"\\])*"', '""', text)
    return re.sub(r"'(\\.|[^'\\])'", "' '", text)


def test_java_sources_are_balanced_and_reference_only_existing_members():
    """No javac here: at least brackets balance, every PinotGpuNative.<member> exists, and every in-package call GpuX.method( resolves to a
    method that class declares with that many parameters."""
    files = java_files()
    native_members = set(re.findall(r"\b(?:int|long|String|void|Object\[\]|long\[\])\s+(\w+)\s*[=(]", strip_c_comments(files["PinotGpuNative.java"])))
    declared = {}
    for fname, text in files.items():
        code = _strip_java_literals(text)
        stack = []
        pairs = {")": "(", "]": "[", "}": "{"}
        for ch in code:
            if ch in "([{":
                stack.append(ch)
            elif ch in ")]}":
                assert stack and stack.pop() == pairs[ch], fname + ": unbalanced " + ch
        assert not stack, fname + ": unclosed " + "".join(stack)
        for member in re.findall(r"PinotGpuNative\.(\w+)", code):
            assert member in native_members, "%s uses PinotGpuNative.%s which does not exist" % (fname, member)
        cls = fname[:-5]
        for name, params in re.findall(r"\b(?:static\s+)?(?:[\w<>\[\], ?]+)\s+(\w+)\s*\(([^)]*)\)\s*(?:throws [\w, .]+)?\s*\{", code):
            declared.setdefault((cls, name), set()).add(0 if not params.strip() else len(re.sub(r"<[^>]*>", "", params).split(",")))
    for fname, text in files.items():
        code = _strip_java_literals(text)
        for cls, name, rest in re.findall(r"\b(Gpu\w+)\.(\w+)\s*\(", code) and [(m.group(1), m.group(2), code[m.end():]) for m in re.finditer(r"\b(Gpu\w+)\.(\w+)\s*\(", code)]:
            if (cls + ".java") not in files or cls == fname[:-5] and False:
                continue
            if name[0].isupper():
                continue                                   # nested class constructor, e.g. GpuAggregationOperator.Lane(
            depth, args, i = 1, 1, 0
            empty = True
            while depth and i < len(rest):
                ch = rest[i]
                if ch in "([{":
                    depth += 1
                elif ch in ")]}":
                    depth -= 1
                elif ch == "," and depth == 1:
                    args += 1
                if depth and not ch.isspace():
                    empty = False
                i += 1
            n = 0 if empty else args
            assert (cls, name) in declared, "%s calls %s.%s which is not declared" % (fname, cls, name)
            assert n in declared[(cls, name)], "%s calls %s.%s with %d arguments, declared with %s" % (fname, cls, name, n, sorted(declared[(cls, name)]))
