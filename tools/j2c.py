#!/usr/bin/env python3
"""j2c.py -- mechanical Java -> C++ transliteration of the reference's codec sources (TEST INFRASTRUCTURE: builds oracle/_ref).

The reference (airlift/aircompressor) is pure Java and this image has no JVM, so the reference's own encoder cannot be *run* here to pin
the oracle's compressed bytes.  What can be done is to compile the reference's own SOURCES: the codec classes are written in a C-like
subset of Java (static methods over sun.misc.Unsafe, int / long arithmetic, arrays, a few small classes), and that subset maps onto
C++ token by token.  This script does exactly that and nothing clever:

  * it reads the .java files named in a manifest from /root/reference WHERE THEY LIE, and writes ONE generated header (into oracle/_ref/,
    which is git-ignored: no reference source text enters the repository);
  * the output keeps the Java line structure -- generated line N of a class is Java line N, `#line` directives name the Java file, so
    compiler diagnostics, debuggers and a reader's diff all point at reference lines;
  * every rule is a token-level rewrite (listed in RULES below); what no rule covers is in the committed patch file
    (oracle/ref/patches.txt: Java file, line range, replacement, reason) -- the audit trail is reference line -> generated line;
  * Java semantics that C++ lacks come from oracle/ref/jrt.h (wrapping arithmetic via -fwrapv, >>> and masked shift counts, array
    handles with bounds checks, Unsafe, string concatenation, exceptions, an arena for `new`).

RULES (token level)
  package / import         dropped (comment); `import static X.NAME` is remembered: an unqualified NAME becomes X::NAME
  modifiers                public / private / protected / final / abstract / ... dropped; `static` kept; annotations dropped; `throws ...` dropped
  class X extends B implements I   ->  struct X : B, I  (bases that are translated or runtime classes; else jobject_base); closing brace gets `;`
  interface                ->  struct with pure virtual methods
  enum                     ->  struct with one static instance per constant (ordinal() as in Java), defined behind the class bodies
  primitive types          byte short int long char boolean float double -> jbyte jshort jint jlong jchar bool jfloat jdouble
  Object / String          jobject (an Unsafe base: null or an array) / jstring
  class types              Foo x -> Foo* x   (objects are references); T[] -> jarray<T>; new T[n] -> jarray<T>::make(n); new T[] {..} -> jarray<T>{..}
  member access            a.b -> a->b when a is an object, A::b when A is a class, a.b otherwise (arrays, strings, UNSAFE)
  unqualified calls        f(..) -> Class::f(..) for a static method of the enclosing / statically imported class, this->f(..) for an instance method
                           (Java keeps methods and variables in separate namespaces; qualification keeps `int hash = hash(x)` legal)
  static fields            primitives: `static inline const T X = ..;` in place; arrays / objects: declared in place, defined behind the class bodies in source order
  literals                 1_000 -> 1000; int hex literals above 0x7FFFFFFF -> (jint)0x..u; "s" -> jstring("s"); null -> nullptr
  shifts                   a >>> b, a >> b, a << b -> a >>JUSHR>> b, a >>JSHR>> b, a <<JSHL<< b (same precedence; jrt.h gives Java's promotion and count masking);
                           x >>>= n etc. -> x = x >>JUSHR>> (n)
  switch                   `case A, B -> stmt` / `-> { .. }` -> `case A: case B: { stmt break; }`; `return switch (..) { case A -> e; }` -> each arm returns
  assert                   -> JASSERT(..) (a no-op: the JVM runs with assertions disabled unless asked)
"""
import os
import re
import sys

REF_MAIN = "/root/reference/src/main/java/io/airlift/compress/v3/"

PRIMS = {"byte": "jbyte", "short": "jshort", "int": "jint", "long": "jlong", "char": "jchar", "boolean": "bool", "float": "jfloat", "double": "jdouble", "void": "void"}
DROP_MODIFIERS = {"public", "private", "protected", "final", "abstract", "transient", "volatile", "synchronized", "strictfp", "native", "sealed"}
# classes the runtime (jrt.h) provides: static access with ::
RUNTIME_STATIC = {"Math", "Integer", "Long", "Short", "Byte", "Arrays", "System", "Objects", "String", "Unsafe"}
RUNTIME_OBJECTS = {"IllegalArgumentException", "IllegalStateException", "UnsupportedOperationException", "IndexOutOfBoundsException", "ArrayIndexOutOfBoundsException",
                   "NullPointerException", "ArithmeticException", "AssertionError", "IOException", "EOFException", "MalformedInputException", "RuntimeException", "Exception",
                   "Throwable", "OutputStream", "InputStream"}
CXX_KEYWORDS = {"register", "union", "struct", "template", "delete", "operator", "signed", "unsigned", "auto", "typename", "namespace", "using", "inline", "extern", "typedef",
                "friend", "explicit", "export", "mutable", "virtual", "and", "or", "not", "xor", "bitand", "bitor", "compl", "near", "far", "errno", "NULL", "EOF"}

TOKEN_RE = re.compile(r"""
    (?P<ws>[ \t\r\n]+)
  | (?P<lc>//[^\n]*)
  | (?P<bc>/\*.*?\*/)
  | (?P<str>"(?:\\.|[^"\\])*")
  | (?P<chr>'(?:\\.|[^'\\])+')
  | (?P<num>(?:0[xX][0-9a-fA-F_]+|0[bB][01_]+|(?:\d[\d_]*\.?[\d_]*|\.\d[\d_]*)(?:[eE][+-]?\d+)?)[lLfFdD]?)
  | (?P<id>[A-Za-z_$][A-Za-z_$0-9]*)
  | (?P<op>>>>=|<<=|>>=|>>>|\+\+|--|->|::|&&|\|\||==|!=|<=|>=|\+=|-=|\*=|/=|%=|&=|\|=|\^=|<<|>>|[{}()\[\];,.@=<>!~?:+\-*/&|^%])
""", re.S | re.X)


class Tok:
    __slots__ = ("kind", "text", "line")

    def __init__(self, kind, text, line):
        self.kind, self.text, self.line = kind, text, line

    def __repr__(self):
        return "%s:%r@%d" % (self.kind, self.text, self.line)


def tokenize(src, fname):
    out, pos, line = [], 0, 1
    while pos < len(src):
        m = TOKEN_RE.match(src, pos)
        if not m:
            raise SystemExit("%s:%d: cannot tokenize %r" % (fname, line, src[pos:pos + 20]))
        kind = m.lastgroup
        text = m.group()
        # `>>` / `>>>` are kept whole: the sources on this path use no nested generics
        out.append(Tok(kind, text, line))
        line += text.count("\n")
        pos = m.end()
    return out


def sig(toks, i, step=1):
    """index of the next significant token from i (inclusive) in direction step, or -1"""
    while 0 <= i < len(toks):
        if toks[i].kind not in ("ws", "lc", "bc"):
            return i
        i += step
    return -1


class ClassInfo:
    def __init__(self, name, outer, kind, fname):
        self.name, self.outer, self.kind, self.fname = name, outer, kind, fname  # kind: class / interface / enum
        self.fields = {}        # name -> (java type string, is_static)
        self.methods = {}       # name -> (java return type string, is_static)   (overloads: all must agree on staticness, checked)
        self.bases = []
        self.enum_constants = []

    def qualified(self):
        return (self.outer.qualified() + "::" if self.outer else "") + self.name


class Project:
    def __init__(self):
        self.classes = {}   # simple name -> ClassInfo
        self.files = []     # (relative path, tokens)

    def is_object_type(self, t):
        """a Java type string naming a reference to an instance of a translated / runtime class (not an array, string, primitive)"""
        return t is not None and not t.endswith("[]") and (t in self.classes or t in RUNTIME_OBJECTS)

    def method_cxx_name(self, cname, mname):
        """Java keeps fields and methods in separate namespaces; a class that has both under one name gets its METHOD renamed (name_m)"""
        seen = set()
        stack = [cname]
        while stack:
            c = stack.pop()
            if c in seen or c not in self.classes:
                continue
            seen.add(c)
            ci = self.classes[c]
            if mname in ci.methods and mname in ci.fields:
                return mname + "_m"
            stack.extend(ci.bases)
        return mname

    def find_field(self, cname, fname):
        seen = set()
        while cname in self.classes and cname not in seen:
            seen.add(cname)
            ci = self.classes[cname]
            if fname in ci.fields:
                return ci.fields[fname]
            nxt = None
            for b in ci.bases:
                if b in self.classes:
                    f = self.find_field(b, fname)
                    if f:
                        return f
            cname = nxt
        return None

    def find_method(self, cname, mname, nargs=None):
        """(java return type, is_static) of method mname of class cname (or a base) taking nargs arguments (any arity if unknown / absent)"""
        if cname in self.classes:
            ci = self.classes[cname]
            if mname in ci.methods:
                by_arity = ci.methods[mname]
                if nargs in by_arity:
                    return by_arity[nargs]
                return next(iter(by_arity.values()))
            for b in ci.bases:
                m = self.find_method(b, mname, nargs)
                if m:
                    return m
        return None


def count_args(toks, open_paren):
    """number of top-level comma-separated items between the parenthesis at open_paren and its match (0 for `()`)"""
    d, n, seen = 0, 0, False
    i = open_paren
    while True:
        x = toks[i]
        if x.kind in ("ws", "lc", "bc"):
            i += 1
            continue
        if x.text in "([{" and x.kind == "op":
            d += 1
            if d > 1:
                seen = True
        elif x.text in ")]}" and x.kind == "op":
            d -= 1
            if d == 0:
                return n + 1 if seen else 0
        elif x.text == "," and d == 1:
            n += 1
        elif d >= 1:
            seen = True
        i += 1


def parse_type(toks, i, proj):
    """a Java type starting at significant token i: base name (possibly Outer.Inner) + [] pairs.  Returns (index after, java type string, base) or None."""
    if toks[i].kind != "id":
        return None
    base = toks[i].text
    if not (base in PRIMS or base in ("Object", "String") or base in proj.classes or base in RUNTIME_OBJECTS):
        return None
    j = i + 1
    # Outer.Inner (nested class named through its outer class)
    while True:
        k = sig(toks, j)
        if k >= 0 and toks[k].text == "." and base in proj.classes:
            k2 = sig(toks, k + 1)
            if k2 >= 0 and toks[k2].kind == "id" and toks[k2].text in proj.classes and proj.classes[toks[k2].text].outer is proj.classes[base]:
                base = toks[k2].text
                j = k2 + 1
                continue
        break
    dims = 0
    while True:
        k = sig(toks, j)
        if k >= 0 and toks[k].text == "[":
            k2 = sig(toks, k + 1)
            if k2 >= 0 and toks[k2].text == "]":
                dims += 1
                j = k2 + 1
                continue
        break
    return j, base + "[]" * dims, base


def cxx_type(jt, proj):
    dims = 0
    while jt.endswith("[]"):
        jt = jt[:-2]
        dims += 1
    if jt in PRIMS:
        t = PRIMS[jt]
    elif jt == "Object":
        t = "jobject"
    elif jt == "String":
        t = "jstring"
    elif jt in proj.classes:
        t = proj.classes[jt].qualified() + "*"
    else:
        t = jt + "*"
    for _ in range(dims):
        t = "jarray<%s>" % t
    return t


# ---------------------------------------------------------------------------------------------------------------------------------
# pass 1: class structure (members and their types)
# ---------------------------------------------------------------------------------------------------------------------------------
def collect(proj, rel, toks):
    stack = []  # (ClassInfo or None, brace depth at which its body opened)
    depth = 0
    parens = 0  # parenthesis depth: method parameters are not members
    i = 0
    n = len(toks)
    pending = None  # ClassInfo whose `{` is awaited
    while i < n:
        t = toks[i]
        if t.kind in ("ws", "lc", "bc", "str", "chr", "num"):
            i += 1
            continue
        if t.kind == "id" and t.text in ("class", "interface", "enum"):
            p = sig(toks, i - 1, -1)
            if p >= 0 and toks[p].text == ".":  # Foo.class
                i += 1
                continue
            k = sig(toks, i + 1)
            name = toks[k].text
            outer = stack[-1][0] if stack else None
            ci = ClassInfo(name, outer, t.text, rel)
            if name in proj.classes:
                raise SystemExit("duplicate class name %s (%s, %s)" % (name, rel, proj.classes[name].fname))
            proj.classes[name] = ci
            # bases up to the `{`
            k = sig(toks, k + 1)
            permits = False
            while toks[k].text != "{":
                permits |= toks[k].text == "permits"  # (sealed interfaces: the permitted subclasses are not bases)
                if toks[k].kind == "id" and toks[k].text not in ("extends", "implements", "permits") and not permits:
                    ci.bases.append(toks[k].text)
                k = sig(toks, k + 1)
            pending = ci
            i = k
            continue
        if t.text == "{":
            depth += 1
            if pending is not None:
                stack.append((pending, depth))
                if pending.kind == "enum":
                    # constants: identifiers at this depth up to `;` or the closing brace
                    k = sig(toks, i + 1)
                    d2 = 0
                    expect = True
                    while k >= 0:
                        x = toks[k]
                        if x.text in "([{":
                            d2 += 1
                        elif x.text in ")]}":
                            if d2 == 0:
                                break
                            d2 -= 1
                        elif d2 == 0 and x.text == ";":
                            break
                        elif d2 == 0 and x.text == ",":
                            expect = True
                        elif d2 == 0 and x.kind == "id" and expect:
                            pending.enum_constants.append(x.text)
                            pending.fields[x.text] = (pending.name, True)
                            expect = False
                        k = sig(toks, k + 1)
                pending = None
            i += 1
            continue
        if t.text == "}":
            if stack and stack[-1][1] == depth:
                stack.pop()
            depth -= 1
            i += 1
            continue
        if t.kind == "op" and t.text in "()":
            parens += 1 if t.text == "(" else -1
            i += 1
            continue
        # a member declaration: at class-body depth, [modifiers] Type name followed by ( or = ; ,
        if stack and stack[-1][1] == depth and parens == 0 and t.kind == "id":
            ci = stack[-1][0]
            # gather modifiers
            j = i
            is_static = False
            while toks[j].kind == "id" and (toks[j].text in DROP_MODIFIERS or toks[j].text in ("static", "default")):
                is_static |= toks[j].text == "static"
                j = sig(toks, j + 1)
            if ci.kind == "interface":
                pass
            pt = parse_type(toks, j, proj) if toks[j].kind == "id" else None
            if pt:
                k = sig(toks, pt[0])
                if k >= 0 and toks[k].kind == "id":
                    k2 = sig(toks, k + 1)
                    if toks[k2].text == "(":
                        nargs = count_args(toks, k2)
                        by_arity = ci.methods.setdefault(toks[k].text, {})
                        prev = by_arity.get(nargs)
                        if prev and prev[1] != is_static:
                            raise SystemExit("%s: overloads of %s with %d parameters differ in staticness" % (rel, toks[k].text, nargs))
                        by_arity[nargs] = (pt[1], is_static)
                    elif toks[k2].text in ("=", ";", ","):
                        ci.fields[toks[k].text] = (pt[1], is_static or ci.kind == "interface")
                        # further declarators: `int a, b;` (no initialisers with commas at this level in the sources; checked below)
                    i = k2
                    # skip to the end of this member header / declaration without descending (the loop handles braces)
                    continue
        i += 1


# ---------------------------------------------------------------------------------------------------------------------------------
# pass 2: rewrite
# ---------------------------------------------------------------------------------------------------------------------------------
class Rewriter:
    def __init__(self, proj, rel, toks, patches):
        self.proj, self.rel, self.toks = proj, rel, toks
        self.out = []           # output text pieces
        self.post = []          # definitions emitted behind the class bodies
        self.post_enum = []     # ... enum constants: ahead of the file's other statics (Java initialises an enum class when its first constant is touched)
        self.static_imports = {}  # name -> owner class
        self.patches = patches  # {first_line: (last_line, text)}
        self.warnings = []

    def err(self, tok, msg):
        raise SystemExit("%s:%d: %s" % (self.rel, tok.line, msg))

    # ---- scopes: name -> java type -------------------------------------------------------------------------------------------
    def lookup_var(self, name):
        for sc in reversed(self.scopes):
            if name in sc:
                return sc[name]
        # fields of the enclosing classes, innermost first
        for ci in reversed(self.class_stack):
            f = self.proj.find_field(ci.name, name)
            if f:
                return f[0]
        return None

    def enclosing_method(self, name, nargs=None):
        for ci in reversed(self.class_stack):
            m = self.proj.find_method(ci.name, name, nargs)
            if m:
                return ci, m
        return None

    def run(self):
        toks, proj = self.toks, self.proj
        n = len(toks)
        out = self.out
        self.scopes = [{}]
        self.class_stack = []
        self.comment_idx = set()
        class_depths = []      # brace depth of each open class body
        depth = 0
        # bracket stack for expression typing: entries (kind, type-after-close)
        brk = []
        last_type = None       # java type of the primary expression just emitted ("class:X" for a class name)
        pending_class = None
        switch_stack = []      # (depth of the switch body, mode)   mode: "stmt" / "ret"
        arrow_close = []       # (depth, text to emit after the matching `}`)
        paren_depth = 0
        i = 0

        def emit(s):
            out.append(s)

        def at_member_level():
            return class_depths and class_depths[-1] == depth

        while i < n:
            t = toks[i]
            # ---- patches: whole line ranges replaced by committed text -----------------------------------------------------------
            if t.line in self.patches and (i == 0 or toks[i - 1].line < t.line or (toks[i - 1].kind in ("ws", "bc", "lc") and "\n" in toks[i - 1].text and toks[i - 1].line + toks[i - 1].text.count("\n") == t.line)):
                last, text, structural = self.patches.pop(t.line)
                j = i
                while j < n and toks[j].line <= last:
                    # keep brace depth in step when the patch replaces braces
                    j += 1
                # the replaced tokens may end inside a whitespace token spanning lines: keep the newlines so that line numbers stay in step
                span_lines = last - t.line + 1
                body = text.rstrip("\n")
                body_lines = body.count("\n") + 1
                emit(body + "\n" * max(span_lines - body_lines, 0))
                # newline that ended the last replaced line
                # (whitespace token holding it was consumed if it started on a replaced line)
                for k in range(i, j):
                    x = toks[k]
                    if x.text == "{":
                        depth += 1
                    elif x.text == "}":
                        depth -= 1
                if j < n and toks[j - 1].kind == "ws" and toks[j - 1].line <= last:
                    tail = toks[j - 1].text
                    # the part of the whitespace after its last newline belongs to the next line's indentation
                    emit("\n" + tail[tail.rfind("\n") + 1:] if "\n" in tail else "")
                last_type = None
                i = j
                continue
            if t.kind in ("ws", "lc", "bc"):
                if t.kind != "ws":
                    self.comment_idx.add(len(out))
                emit(t.text)
                i += 1
                continue
            if t.kind == "str":
                emit("jstring(" + t.text + ")")
                last_type = "String"
                i += 1
                continue
            if t.kind == "chr":
                emit(t.text)
                last_type = "char"
                i += 1
                continue
            if t.kind == "num":
                emit(self.number(t))
                last_type = None
                i += 1
                continue
            nx = sig(toks, i + 1)
            nxt = toks[nx].text if nx >= 0 else ""
            pv = sig(toks, i - 1, -1)
            prev = toks[pv].text if pv >= 0 else ""
            if t.kind == "id":
                w = t.text
                # ---- package / import ------------------------------------------------------------------------------------------
                if w in ("package", "import") and depth == 0:
                    j = i
                    words = []
                    while toks[j].text != ";":
                        if toks[j].kind == "id":
                            words.append(toks[j].text)
                        j += 1
                    if w == "import" and len(words) > 2 and words[1] == "static":
                        self.static_imports[words[-1]] = words[-2]
                    emit("// " + "".join(x.text for x in toks[i:j + 1]))
                    i = j + 1
                    continue
                if w == "@":  # (never an id; kept for clarity)
                    pass
                # ---- modifiers ---------------------------------------------------------------------------------------------------
                if w in DROP_MODIFIERS or w == "default" and at_member_level() and nxt != ":":
                    # swallow the following whitespace too (keeps columns tidy, never a newline)
                    i += 1
                    if i < n and toks[i].kind == "ws" and "\n" not in toks[i].text:
                        i += 1
                    continue
                if w == "throws":
                    j = nx
                    while toks[j].text not in ("{", ";"):
                        j += 1
                    # drop `throws A, B` and the whitespace before it
                    while out and out[-1].strip() == "" and "\n" not in out[-1]:
                        out.pop()
                    ws = "".join(x.text for x in toks[i:j] if x.kind == "ws" and "\n" in x.text)
                    emit(ws if ws else " ")
                    i = j
                    continue
                # ---- class / interface / enum headers -----------------------------------------------------------------------
                if w in ("class", "interface", "enum") and prev != ".":
                    name = toks[nx].text
                    ci = proj.classes[name]
                    j = sig(toks, nx + 1)
                    bases = []
                    hdr_ws = ""
                    permits = False
                    while toks[j].text != "{":
                        permits |= toks[j].text == "permits"
                        if toks[j].kind == "id" and toks[j].text not in ("extends", "implements", "permits") and not permits:
                            b = toks[j].text
                            if b in proj.classes:
                                bases.append(proj.classes[b].qualified())
                            elif b in RUNTIME_OBJECTS:
                                bases.append(b)
                        j += 1
                    hdr_ws = "".join(x.text for x in toks[nx + 1:j] if x.kind == "ws" and "\n" in x.text)
                    needs_root = not any(True for b in ci.bases if (b in proj.classes and proj.classes[b].kind != "interface") or b in RUNTIME_OBJECTS)
                    if ci.kind == "interface":
                        base_list = bases
                    else:
                        base_list = (["jobject_base"] if needs_root and ci.kind != "enum" else []) + (["jenum_base"] if ci.kind == "enum" else []) + bases
                    # `static class` inside a class: the `static` has been emitted already -- take it back
                    k = len(out) - 1
                    while k >= 0 and out[k].strip() == "":
                        k -= 1
                    if k >= 0 and out[k] == "static":
                        del out[k:]
                    emit("struct " + name + (" : " + ", ".join(base_list) if base_list else "") + (hdr_ws if hdr_ws else " "))
                    pending_class = ci
                    i = j
                    continue
                # ---- assert ---------------------------------------------------------------------------------------------------------
                if w == "assert":
                    j = nx
                    d2 = 0
                    cond_end = None
                    while True:
                        x = toks[j].text
                        if x in "([":
                            d2 += 1
                        elif x in ")]":
                            d2 -= 1
                        elif d2 == 0 and x == "?":
                            d2 += 100  # a conditional expression: its `:` is not the message separator
                        elif d2 >= 100 and x == ":":
                            d2 -= 100
                        elif d2 == 0 and x == ":" and cond_end is None:
                            cond_end = j
                        elif d2 == 0 and x == ";":
                            break
                        j += 1
                    text = "".join(x.text for x in toks[nx:(cond_end if cond_end is not None else j)])
                    emit("JASSERT(" + re.sub(r">>>", ">>JUSHR>>", text).rstrip() + ");")
                    i = j + 1
                    continue
                # ---- null / literals-as-words -------------------------------------------------------------------------------------
                if w == "null":
                    emit("nullptr")
                    last_type = None
                    i += 1
                    continue
                if w == "this":
                    emit("this")
                    last_type = self.class_stack[-1].name if self.class_stack else None
                    i += 1
                    continue
                if w == "instanceof":
                    self.err(t, "instanceof: needs a patch")
                # ---- new -------------------------------------------------------------------------------------------------------------
                if w == "new":
                    pt = parse_type_after_new(toks, nx, proj)
                    if pt is None:
                        self.err(t, "new of an unknown type %s" % toks[nx].text)
                    j, base, dims_exprs, empty_dims, has_init = pt
                    if dims_exprs or empty_dims:
                        elem = base
                        total = len(dims_exprs) + empty_dims
                        ct = cxx_type(elem + "[]" * total, proj)
                        if has_init:
                            emit(ct)  # followed by the `{ ... }` initialiser, which is emitted as it comes
                            last_type = None
                        else:
                            if len(dims_exprs) != 1:
                                self.err(t, "multi-dimensional `new` with sizes: needs a patch")
                            emit(ct + "::make(")
                            emit(self.fragment(dims_exprs[0]))
                            emit(")")
                            last_type = elem + "[]" * total
                        i = j
                        continue
                    # new C(args): the constructor call is emitted as it comes; the value is an object of type base
                    emit("new " + (proj.classes[base].qualified() if base in proj.classes else base))
                    k = sig(toks, j)
                    if toks[k].text != "(":
                        self.err(t, "new %s without arguments" % base)
                    brk.append(("call", base))
                    emit("".join(x.text for x in toks[j:k]) + "(")
                    paren_depth += 1
                    last_type = None
                    i = k + 1
                    continue
                # ---- switch ---------------------------------------------------------------------------------------------------------
                if w == "switch":
                    mode = "ret" if prev == "return" else "stmt"
                    if mode == "ret":
                        # take the `return` back: every arm returns
                        k = len(out) - 1
                        while out[k].strip() == "":
                            k -= 1
                        assert out[k] == "return", out[k]
                        del out[k:]
                    self.pending_switch = mode
                    emit("switch")
                    i += 1
                    continue
                if w in ("case", "default") and switch_stack and (w == "case" or nxt in ("->", ":")):
                    # find the end of the labels: `->` (arrow form) or `:` (classic)
                    j = nx
                    d2 = 0
                    while True:
                        x = toks[j].text
                        if x in "([":
                            d2 += 1
                        elif x in ")]":
                            d2 -= 1
                        elif d2 == 0 and x in ("->", ":"):
                            break
                        elif d2 == 0 and x == "?":
                            self.err(t, "conditional expression in a case label")
                        j += 1
                    arrow = toks[j].text == "->"
                    if w == "default":
                        emit("default:")
                    else:
                        # labels separated by commas at depth 0
                        labels, cur, d2 = [], [], 0
                        for x in toks[nx:j]:
                            if x.text in "([":
                                d2 += 1
                            elif x.text in ")]":
                                d2 -= 1
                            if d2 == 0 and x.text == ",":
                                labels.append(cur)
                                cur = []
                            else:
                                cur.append(x)
                        labels.append(cur)
                        parts = []
                        for lab in labels:
                            parts.append("case " + self.fragment(lab).strip() + ":")
                        emit(" ".join(parts))
                    if arrow:
                        mode = switch_stack[-1][1]
                        k = sig(toks, j + 1)
                        if toks[k].text == "{":
                            # block arm: `{ ... }` then break (statement switches only)
                            if mode == "ret":
                                self.err(t, "block arm in a switch expression: needs a patch")
                            arrow_close.append((depth + 1, " break;"))
                        elif toks[k].text == "throw":
                            pass  # no break behind a throw
                        else:
                            # expression / statement arm up to its `;`
                            if mode == "ret":
                                emit(" return")
                            else:
                                self.arm_break_at = self.statement_end(k)
                    i = j + 1
                    last_type = None
                    continue
                # ---- declarations and casts: a type in type position ----------------------------------------------------------------
                pt = parse_type(toks, i, proj)
                if pt is not None and prev not in (".",):
                    j, jt, base = pt
                    k = sig(toks, j)
                    after = toks[k] if k >= 0 else None
                    is_decl = after is not None and after.kind == "id" and after.text not in ("instanceof",) and not (base in proj.classes and jt == base and False)
                    is_cast = prev == "(" and after is not None and after.text == ")" and (base in PRIMS or self.cast_follows(k))
                    is_class_literal = False
                    if is_decl:
                        vname = after.text
                        k2 = sig(toks, k + 1)
                        follows = toks[k2].text if k2 >= 0 else ""
                        if follows == "(":
                            # a method header (return type + name): open a scope for its parameters
                            ci = self.class_stack[-1]
                            static_kw = self.take_back_static()
                            virt = "virtual " if (ci.kind == "interface" or (not static_kw)) and ci.kind != "enum" or (ci.kind == "enum" and not static_kw) else ""
                            emit(("static " if static_kw else virt) + cxx_type(jt, proj))
                            emit("".join(x.text for x in toks[j:k]) + self.safe_name(proj.method_cxx_name(ci.name, vname)))
                            self.method_pending = (ci, k2)
                            self.scopes.append({})
                            self.method_scope_depth = depth
                            i = k + 1
                            last_type = None
                            continue
                        # a variable / field / parameter
                        self.scopes[-1][vname] = jt
                        if at_member_level() and paren_depth == 0:
                            ci = self.class_stack[-1]
                            static_kw = self.take_back_static() or ci.kind == "interface"
                            if static_kw:
                                # static field: primitive -> inline const in place; everything else declared here, defined behind the classes
                                end = self.statement_end(k)
                                has_init = follows == "="
                                if base in PRIMS and not jt.endswith("[]"):
                                    emit("static inline const " + cxx_type(jt, proj))
                                    emit("".join(x.text for x in toks[j:k]) + self.safe_name(vname))
                                    i = k + 1
                                    continue
                                emit("static " + cxx_type(jt, proj) + "".join(x.text for x in toks[j:k]) + self.safe_name(vname) + ";")
                                if has_init:
                                    init_toks = toks[k2 + 1:end]
                                    init = self.fragment(init_toks)
                                    q = ci.qualified()
                                    self.post.append("#line %d \"%s\"\n%s Ref::%s::%s =%s;\n" % (t.line, self.rel, cxx_type_global(jt, proj), q, self.safe_name(vname), init))
                                    # keep the line structure: the initialiser's newlines
                                    emit("\n" * sum(x.text.count("\n") for x in toks[k2:end]))
                                i = end + 1
                                last_type = None
                                continue
                        emit(cxx_type(jt, proj))
                        emit("".join(x.text for x in toks[j:k]))
                        i = k
                        last_type = None
                        continue
                    if is_cast:
                        emit(cxx_type(jt, proj))
                        i = j
                        last_type = None
                        continue
                    if jt != base and False:
                        pass
                    # for-each / catch handled as declarations above; otherwise the name is used as a value (static access)
                # ---- identifiers in expressions -----------------------------------------------------------------------------------
                name = self.safe_name(w)
                if prev == ".":
                    # member of last_type
                    owner = last_type
                    if owner and owner.startswith("class:"):
                        owner = owner[6:]
                    if nxt == "(":
                        m = proj.find_method(owner, w, count_args(toks, nx)) if owner else None
                        brk.append(("call", m[0] if m else None))
                        emit(self.safe_name(proj.method_cxx_name(owner, w)) if owner else name)
                        emit("".join(x.text for x in toks[i + 1:nx]) + "(")
                        paren_depth += 1
                        i = nx + 1
                        last_type = None
                        continue
                    f = proj.find_field(owner, w) if owner else None
                    if owner in proj.classes and w in proj.classes and proj.classes[w].outer is proj.classes[owner]:
                        last_type = "class:" + w
                    elif w == "length" and owner and owner.endswith("[]"):
                        last_type = "int"
                    else:
                        last_type = f[0] if f else None
                    emit(name)
                    i += 1
                    continue
                if nxt == "(" and w not in ("if", "while", "for", "switch", "return", "catch", "synchronized", "super", "throw", "else", "do", "try"):
                    # an unqualified call
                    ret = None
                    if self.class_stack and w == self.class_stack[-1].name and at_member_level():
                        # constructor header
                        self.take_back_static()
                        emit(name)
                        self.method_pending = (self.class_stack[-1], nx)
                        self.scopes.append({})
                        self.method_scope_depth = depth
                        i += 1
                        continue
                    em = self.enclosing_method(w, count_args(toks, nx))
                    if em:
                        ci, (ret, is_static) = em
                        emit((ci.qualified() + "::" if is_static else "this->") + self.safe_name(proj.method_cxx_name(ci.name, w)))
                    elif w in self.static_imports:
                        owner = self.static_imports[w]
                        m = proj.find_method(owner, w, count_args(toks, nx))
                        ret = m[0] if m else None
                        emit((proj.classes[owner].qualified() if owner in proj.classes else owner) + "::" + self.safe_name(proj.method_cxx_name(owner, w)))
                    else:
                        emit(name)
                    brk.append(("call", ret))
                    emit("".join(x.text for x in toks[i + 1:nx]) + "(")
                    paren_depth += 1
                    i = nx + 1
                    last_type = None
                    continue
                vt = self.lookup_var(w)
                if vt is not None:
                    # a field of an enclosing class that is static and not this class's: qualify? (same class scope in C++: nested classes see outer statics)
                    emit(name)
                    last_type = vt
                elif w in self.static_imports:
                    owner = self.static_imports[w]
                    if w == "UNSAFE":
                        emit("UNSAFE")
                        last_type = "Unsafe"
                    elif owner == "Unsafe":
                        emit("Unsafe::" + name)
                        last_type = "int"
                    else:
                        f = proj.find_field(owner, w)
                        emit((proj.classes[owner].qualified() if owner in proj.classes else owner) + "::" + name)
                        last_type = f[0] if f else None
                elif w in proj.classes:
                    emit(proj.classes[w].qualified() if nxt != "." else name)
                    last_type = "class:" + w
                elif w in RUNTIME_STATIC or w in RUNTIME_OBJECTS:
                    emit(name)
                    last_type = "class:" + w
                else:
                    emit(name)
                    last_type = None
                i += 1
                continue
            # ---- operators and punctuation ------------------------------------------------------------------------------------------
            x = t.text
            if x == "@":
                # annotation: @Name or @Name(...)
                j = sig(toks, nx + 1)
                if j >= 0 and toks[j].text == "(":
                    d2 = 0
                    while True:
                        if toks[j].text == "(":
                            d2 += 1
                        elif toks[j].text == ")":
                            d2 -= 1
                            if d2 == 0:
                                break
                        j += 1
                    i = j + 1
                else:
                    i = nx + 1
                continue
            if x == ".":
                lt = last_type
                if lt and lt.startswith("class:"):
                    emit("::")
                elif lt == "Unsafe" or lt is None and prev == "UNSAFE":
                    emit(".")
                elif proj.is_object_type(lt):
                    emit("->")
                else:
                    emit(".")
                i += 1
                continue  # (last_type stays: the member lookup needs it)
            if x in (">>>", ">>", "<<"):
                emit({">>>": ">>JUSHR>>", ">>": ">>JSHR>>", "<<": "<<JSHL<<"}[x])
                last_type = None
                i += 1
                continue
            if x in (">>>=", ">>=", "<<="):
                # lhs = lhs OP (rhs);  -- lhs is everything emitted since the statement began
                k = len(out) - 1
                lhs = []
                while k >= 0 and (k in self.comment_idx or not out[k].rstrip().endswith((";", "{", "}"))):
                    if k not in self.comment_idx:
                        lhs.insert(0, out[k])
                    k -= 1
                lhs_text = "".join(lhs).strip()
                if not re.match(r"^[A-Za-z_][\w\.\->\[\]]*$", lhs_text):
                    self.err(t, "compound shift on a complex left side (%r): needs a patch" % lhs_text)
                end = self.statement_end(i)
                rhs = self.fragment(toks[i + 1:end]).strip()
                op = {">>>=": ">>JUSHR>>", ">>=": ">>JSHR>>", "<<=": "<<JSHL<<"}[x]
                emit("= " + lhs_text + " " + op + " (" + rhs + ")")
                i = end
                last_type = None
                continue
            if x == "{":
                depth += 1
                emit("{")
                if pending_class is not None:
                    ci = pending_class
                    pending_class = None
                    self.class_stack.append(ci)
                    class_depths.append(depth)
                    self.scopes.append({})
                    # Java lets a class use its nested classes before their definition: declare them up front
                    nested = [c.name for c in proj.classes.values() if c.outer is ci]
                    if nested:
                        emit(" " + " ".join("struct %s;" % x for x in nested))
                    if ci.kind == "enum":
                        i = self.enum_constants(ci, i + 1)
                        continue
                elif getattr(self, "pending_switch", None) and prev == ")":
                    switch_stack.append((depth, self.pending_switch))
                    self.pending_switch = None
                elif getattr(self, "method_pending", None):
                    self.method_pending = None
                    self.method_body_depth = depth
                else:
                    self.scopes.append({})
                    self.block_scopes = getattr(self, "block_scopes", []) + [depth]
                last_type = None
                i += 1
                continue
            if x == "}":
                emit("}")
                if class_depths and class_depths[-1] == depth:
                    class_depths.pop()
                    self.class_stack.pop()
                    self.scopes.pop()
                    emit(";")
                elif switch_stack and switch_stack[-1][0] == depth:
                    switch_stack.pop()
                elif getattr(self, "method_body_depth", None) == depth and getattr(self, "method_scope_depth", None) == depth - 1:
                    self.scopes.pop()
                    self.method_body_depth = None
                elif getattr(self, "block_scopes", None) and self.block_scopes[-1] == depth:
                    self.block_scopes.pop()
                    self.scopes.pop()
                if arrow_close and arrow_close[-1][0] == depth:
                    emit(arrow_close.pop()[1])
                depth -= 1
                last_type = None
                i += 1
                continue
            if x == "(":
                brk.append(("paren", None))
                paren_depth += 1
                emit("(")
                last_type = None
                i += 1
                continue
            if x == ")":
                kind, ty = brk.pop() if brk else ("paren", None)
                paren_depth -= 1
                emit(")")
                last_type = ty if kind == "call" else last_type
                i += 1
                continue
            if x == "[":
                et = last_type[:-2] if last_type and last_type.endswith("[]") else None
                brk.append(("index", et))
                emit("[")
                last_type = None
                i += 1
                continue
            if x == "]":
                kind, ty = brk.pop() if brk else ("index", None)
                emit("]")
                last_type = ty
                i += 1
                continue
            if x == ";":
                emit(";")
                if getattr(self, "method_pending", None) and paren_depth == 0:
                    # a method without a body (interface): pure virtual
                    ci, _ = self.method_pending
                    out.pop()
                    emit(" = 0;")
                    self.method_pending = None
                    self.scopes.pop()
                if getattr(self, "arm_break_at", None) == i:
                    emit(" break;")
                    self.arm_break_at = None
                last_type = None
                i += 1
                continue
            if x == "->":
                self.err(t, "lambda / unexpected arrow: needs a patch")
            emit(x)
            last_type = None
            i += 1
        return "".join(out)

    def fragment(self, toks):
        """rewrite a token slice (an expression) in this rewriter's context: scopes, enclosing classes, static imports"""
        return FragmentRewriter(self.proj, self.rel, toks, self.scopes, self.class_stack, self.static_imports).run()

    # ---- helpers ---------------------------------------------------------------------------------------------------------------------
    def number(self, t):
        s = t.text.replace("_", "")
        m = re.match(r"^(0[xX][0-9a-fA-F]+|0[bB][01]+)([lL]?)$", s)
        if m and not m.group(2):
            v = int(m.group(1), 0)
            if v > 0x7FFFFFFF:
                return "((jint)%su)" % m.group(1)
        if re.match(r"^\d+[lL]$", s) or (m and m.group(2)):
            return s[:-1] + "L"  # (long == jlong on LP64)
        if re.match(r"^0\d+$", s):
            self.err(t, "octal literal")
        if m:
            return s  # (hex / binary: a trailing d or f is a digit)
        if s[-1] in "dD":
            return s[:-1]
        return s

    def safe_name(self, w):
        return w + "_" if w in CXX_KEYWORDS else w

    def statement_end(self, i):
        """index of the `;` that ends the statement token i belongs to (brackets balanced)"""
        toks = self.toks
        d = 0
        while i < len(toks):
            x = toks[i].text
            if toks[i].kind in ("str", "chr"):
                i += 1
                continue
            if x in "([{":
                d += 1
            elif x in ")]}":
                d -= 1
            elif x == ";" and d == 0:
                return i
            i += 1
        raise SystemExit("%s: statement without end" % self.rel)

    def cast_follows(self, close_paren):
        """`(Type)` followed by something a cast applies to"""
        k = sig(self.toks, close_paren + 1)
        if k < 0:
            return False
        x = self.toks[k]
        return x.kind in ("id", "num", "str", "chr") or x.text in ("(", "~", "!", "-", "+")

    def take_back_static(self):
        """if the last significant output piece is `static`, remove it (the caller re-emits it in its place) and say so"""
        out = self.out
        k = len(out) - 1
        while k >= 0 and out[k].strip() == "":
            k -= 1
        if k >= 0 and out[k] == "static":
            del out[k:]
            return True
        return False

    def enum_constants(self, ci, i):
        """the constant list of an enum body, starting behind its `{`: emits the static instance declarations; returns the index to continue at"""
        toks = self.toks
        consts = []  # (name, args tokens or None, line)
        k = i
        d = 0
        cur = None
        args = None
        start = i
        while True:
            x = toks[k]
            if x.kind in ("ws", "lc", "bc"):
                k += 1
                continue
            if x.text == "(":
                if d == 0:
                    args = []
                else:
                    args.append(x)
                d += 1
            elif x.text == ")":
                d -= 1
                if d > 0:
                    args.append(x)
            elif d > 0:
                args.append(x)
            elif x.kind == "id":
                cur = (x.text, x.line)
            elif x.text in (",", ";", "}"):
                if cur:
                    consts.append((cur[0], args, cur[1]))
                cur, args = None, None
                if x.text != ",":
                    break
            k += 1
        end = k
        newlines = sum(t.text.count("\n") for t in toks[start:end])
        decls = " ".join("static %s* %s;" % (ci.name, self.safe_name(c[0])) for c in consts)
        self.out.append(" " + decls + "\n" * newlines)
        q = ci.qualified()
        for ordinal, (name, args, line) in enumerate(consts):
            a = ""
            if args:
                a = FragmentRewriter(self.proj, self.rel, args, self.scopes, self.class_stack, self.static_imports).run()
            self.post_enum.append("#line %d \"%s\"\nRef::%s* Ref::%s::%s = jenum_make(new Ref::%s(%s), %d, \"%s\");\n" % (line, self.rel, q, q, self.safe_name(name), q, a, ordinal, name))
        return end + (1 if toks[end].text == ";" else 0)


class FragmentRewriter(Rewriter):
    """an expression fragment rewritten in the context (scopes, enclosing classes, static imports) of its parent"""

    def __init__(self, proj, rel, toks, scopes, class_stack, static_imports):
        Rewriter.__init__(self, proj, rel, toks, {})
        self._ctx = (scopes, class_stack)
        self.static_imports = static_imports

    def run(self):
        # Rewriter.run resets scopes / class_stack first; re-seed right after by wrapping the attributes as properties would be overkill:
        # run the parent's loop on a copy whose reset values are the context
        scopes, class_stack = self._ctx
        self._reset_scopes = list(scopes)
        self._reset_classes = list(class_stack)
        return Rewriter.run(self)

    def __setattr__(self, k, v):
        if k == "scopes" and hasattr(self, "_reset_scopes") and v == [{}]:
            v = self._reset_scopes + [{}]
        if k == "class_stack" and hasattr(self, "_reset_classes") and v == []:
            v = list(self._reset_classes)
        object.__setattr__(self, k, v)


def parse_type_after_new(toks, i, proj):
    """after `new`: Type [expr]... or Type [] ... { init } or Type ( args ).  Returns (index after the type part, base, [dim expr token lists], empty dims, has initialiser)"""
    if toks[i].kind != "id":
        return None
    base = toks[i].text
    j = i + 1
    while True:
        k = sig(toks, j)
        if toks[k].text == "." and base in proj.classes:
            k2 = sig(toks, k + 1)
            if toks[k2].kind == "id" and toks[k2].text in proj.classes:
                base = toks[k2].text
                j = k2 + 1
                continue
        break
    if not (base in PRIMS or base in proj.classes or base in RUNTIME_OBJECTS or base in ("Object", "String")):
        return None
    dims, empty = [], 0
    while True:
        k = sig(toks, j)
        if toks[k].text != "[":
            break
        k2 = sig(toks, k + 1)
        if toks[k2].text == "]":
            empty += 1
            j = k2 + 1
            continue
        # a sized dimension: tokens up to the matching ]
        d = 0
        e = k + 1
        while True:
            if toks[e].text == "[":
                d += 1
            elif toks[e].text == "]":
                if d == 0:
                    break
                d -= 1
            e += 1
        dims.append(toks[k + 1:e])
        j = e + 1
    k = sig(toks, j)
    has_init = (dims or empty) and toks[k].text == "{"
    return j, base, dims, empty, bool(has_init)


def cxx_type_global(jt, proj):
    """cxx_type with class names qualified from outside struct Ref"""
    dims = 0
    while jt.endswith("[]"):
        jt = jt[:-2]
        dims += 1
    if jt in proj.classes:
        t = "Ref::" + proj.classes[jt].qualified() + "*"
    else:
        t = cxx_type(jt, proj)
    for _ in range(dims):
        t = "jarray<%s>" % t
    return t


def load_patches(path):
    """patch file format: blocks
         @@ <java file relative to M/> <first line>[-<last line>]  # reason
         replacement text (C++), any number of lines
       The replaced Java lines are shown by `j2c.py --audit`."""
    patches = {}
    if not os.path.exists(path):
        return patches
    cur = None
    for raw in open(path):
        if raw.startswith("@@post "):
            # text for the section behind the class bodies (static definitions a patch declares), emitted where the file's own entries go
            cur = [0, "", True]
            patches.setdefault(raw.split()[1], {}).setdefault("post", []).append(cur)
        elif raw.startswith("@@ "):
            m = re.match(r"@@ (\S+) (\d+)(?:-(\d+))?", raw)
            f, a, b = m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))
            cur = [b, "", True]
            patches.setdefault(f, {})[a] = cur
        elif raw.startswith("#@") or cur is None:
            continue
        else:
            cur[1] += raw
    return {f: {a: (tuple(v) if a != "post" else "".join(x[1] for x in v)) for a, v in d.items()} for f, d in patches.items()}


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--manifest", required=True, help="file listing the Java sources (relative to %s), in emission order" % REF_MAIN)
    ap.add_argument("--patches", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--root", default=REF_MAIN)
    ap.add_argument("--outer", default="Ref", help="name of the struct that encloses the translated classes (one per generated header)")
    args = ap.parse_args()
    rels = [l.split("#")[0].strip() for l in open(args.manifest)]
    rels = [r for r in rels if r]
    # `@runtime Name`: a class the recipe provides by hand (oracle/ref/ref_extra.h); static access Name::member
    for r in rels:
        if r.startswith("@runtime "):
            RUNTIME_STATIC.add(r.split()[1])
    rels = [r for r in rels if not r.startswith("@")]
    proj = Project()
    srcs = {}
    for rel in rels:
        src = open(os.path.join(args.root, rel)).read()
        toks = tokenize(src, rel)
        srcs[rel] = toks
    for rel in rels:
        collect(proj, rel, srcs[rel])
    patches = load_patches(args.patches)
    body, post = [], []
    fwd = []
    for name, ci in proj.classes.items():
        if ci.outer is None:
            fwd.append("    struct %s;" % name)
    for rel in rels:
        file_patches = dict(patches.get(rel, {}))
        post_patch = file_patches.pop("post", "")
        rw = Rewriter(proj, rel, srcs[rel], file_patches)
        text = rw.run()
        if post_patch:
            post.append("// (oracle/ref/patches.txt, @@post %s)\n%s" % (rel, post_patch))
        if rw.patches:
            raise SystemExit("%s: patches not applied at lines %s" % (rel, sorted(rw.patches)))
        body.append("#line 1 \"%s\"\n%s\n" % ("M/" + rel, text))
        post.extend(p.replace('"%s"' % rel, '"M/%s"' % rel) for p in rw.post_enum + rw.post)
    with open(args.out, "w") as f:
        f.write("// GENERATED by tools/j2c.py from the reference's Java sources (see oracle/ref/README.md) -- not committed, not linked into the product.\n")
        f.write("#pragma once\n#include \"jrt.h\"\n#include \"ref_extra.h\"\nstruct %s {\n" % args.outer + "\n".join(fwd) + "\n")
        f.write("".join(body))
        f.write("};\n// ---- static fields of reference type and enum constants, in source order ----\n")
        f.write(re.sub(r"\bRef(::|_)", args.outer + r"\1", "".join(post)))
    print("j2c: %d files, %d classes -> %s" % (len(rels), len(proj.classes), args.out))


if __name__ == "__main__":
    main()
