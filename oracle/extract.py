#!/usr/bin/env python3
"""Pull named top-level items out of a reference C source file, verbatim.

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Used by oracle/Makefile to
build oracle/_ref/ from the reference sources *where they lie* under
/root/reference: whole iop translation units cannot be compiled here (they
include GTK headers), so the build recipe lifts the pixel functions it needs
out of them, unchanged, into a scratch directory and compiles that.  Nothing
this script emits is ever committed.

usage: extract.py SRC OUT NAME [NAME...]

An item is a top-level function definition, typedef, enum/struct definition,
static-const object, or a #define.  Items are emitted in source order.
"""
import re
import sys


def _scan_chunks(text):
    """Split C source into top-level chunks: (start, end, kind) with kind in
    {'pp', 'decl'}.  Comments, strings and char literals are skipped when
    tracking brace/paren depth."""
    n = len(text)
    i = 0
    chunks = []
    start = None          # start of the current declaration chunk
    depth_brace = 0
    depth_paren = 0
    saw_brace_close_at0 = False
    while i < n:
        c = text[i]
        # comments
        if c == '/' and i + 1 < n and text[i + 1] == '/':
            j = text.find('\n', i)
            i = n if j < 0 else j
            continue
        if c == '/' and i + 1 < n and text[i + 1] == '*':
            j = text.find('*/', i + 2)
            i = n if j < 0 else j + 2
            continue
        if c == '"' or c == "'":
            q = c
            i += 1
            while i < n and text[i] != q:
                if text[i] == '\\':
                    i += 1
                i += 1
            i += 1
            continue
        # preprocessor line at top level
        if c == '#' and depth_brace == 0 and depth_paren == 0:
            # must be first non-blank on the line
            ls = text.rfind('\n', 0, i) + 1
            if text[ls:i].strip() == '':
                if start is not None:
                    # a macro invocation without ';' (DT_MODULE_INTROSPECTION(...)) is pending:
                    # close it here so the directive is seen on its own
                    chunks.append((start, ls, 'decl'))
                    start = None
                j = i
                while True:
                    e = text.find('\n', j)
                    if e < 0:
                        e = n
                        break
                    if text[e - 1] == '\\':
                        j = e + 1
                        continue
                    break
                chunks.append((i, e, 'pp'))
                i = e
                continue
        if c.isspace():
            i += 1
            continue
        if start is None:
            start = i
            saw_brace_close_at0 = False
        if c == '(':
            depth_paren += 1
        elif c == ')':
            depth_paren -= 1
        elif c == '{':
            depth_brace += 1
        elif c == '}':
            depth_brace -= 1
            if depth_brace == 0 and depth_paren == 0:
                # function body end, or struct/enum/initializer end (then ';' follows)
                j = i + 1
                while j < n and text[j] in ' \t':
                    j += 1
                # look ahead: if the next significant char sequence leads to ';' before
                # any other declaration start, this is a struct/typedef/initializer
                k = j
                # skip whitespace/newlines
                while k < n and text[k].isspace():
                    k += 1
                head = text[start:i + 1]
                is_func = _looks_like_function(head)
                if is_func:
                    chunks.append((start, i + 1, 'decl'))
                    start = None
                    i += 1
                    continue
        elif c == ';' and depth_brace == 0 and depth_paren == 0:
            chunks.append((start, i + 1, 'decl'))
            start = None
        i += 1
    return chunks


def _strip_comments(s):
    s = re.sub(r'/\*.*?\*/', ' ', s, flags=re.S)
    s = re.sub(r'//[^\n]*', ' ', s)
    return s


def _looks_like_function(head):
    """head = text from chunk start up to and including the '}' closing at depth 0."""
    s = _strip_comments(head)
    b = _first_brace_at_paren0(s)
    if b < 0:
        return False
    pre = s[:b].rstrip()
    if not pre.endswith(')'):
        return False
    if re.match(r'\s*typedef\b', s):
        return False
    if '=' in _outside_parens(pre):
        return False
    return True


def _outside_parens(s):
    out = []
    d = 0
    for ch in s:
        if ch == '(':
            d += 1
        elif ch == ')':
            d -= 1
        elif d == 0:
            out.append(ch)
    return ''.join(out)


def _first_brace_at_paren0(s):
    d = 0
    for i, ch in enumerate(s):
        if ch == '(':
            d += 1
        elif ch == ')':
            d -= 1
        elif ch == '{' and d == 0:
            return i
    return -1


def _chunk_names(text, kind):
    s = _strip_comments(text)
    if kind == 'pp':
        m = re.match(r'\s*#\s*define\s+(\w+)', s)
        return [m.group(1)] if m else []
    names = []
    if _looks_like_function(s):
        b = _first_brace_at_paren0(s)
        pre = s[:b].rstrip()
        # walk back over the parameter list
        d = 0
        i = len(pre) - 1
        while i >= 0:
            if pre[i] == ')':
                d += 1
            elif pre[i] == '(':
                d -= 1
                if d == 0:
                    break
            i -= 1
        m = re.search(r'(\w+)\s*$', pre[:i])
        if m:
            names.append(m.group(1))
        return names
    # typedef / struct / enum / object
    m = re.match(r'\s*typedef\b', s)
    if m:
        # function-pointer typedef: (*name)
        fp = re.search(r'\(\s*\*\s*(\w+)\s*\)', _strip_braces(s))
        body = _strip_braces(s).rstrip().rstrip(';').rstrip()
        # array typedef: name[...]
        body2 = re.sub(r'(\[[^\]]*\])+\s*$', '', body)
        body2 = re.sub(r'__attribute__\s*\(\(.*?\)\)\s*$', '', body2).rstrip()
        mm = re.search(r'(\w+)\s*$', body2)
        if fp:
            names.append(fp.group(1))
        if mm:
            names.append(mm.group(1))
        tag = re.match(r'\s*typedef\s+(?:struct|enum|union)\s+(\w+)', s)
        if tag:
            names.append(tag.group(1))
        return names
    m = re.match(r'\s*(?:struct|enum|union)\s+(\w+)\s*\{', s)
    if m:
        names.append(m.group(1))
        return names
    # object definition: take identifier before first '[' or '=' at depth 0
    flat = _strip_braces(s)
    head = re.split(r'[=\[;]', flat, 1)[0]
    head = re.sub(r'__attribute__\s*\(\(.*?\)\)', ' ', head)
    head = re.sub(r'\bDT_ALIGNED_(PIXEL|ARRAY)\b', ' ', head)
    mm = re.search(r'(\w+)\s*$', head.rstrip())
    if mm:
        names.append(mm.group(1))
    # enumerators of anonymous enums are not indexed
    return names


def _strip_braces(s):
    out = []
    d = 0
    for ch in s:
        if ch == '{':
            d += 1
        elif ch == '}':
            d -= 1
        elif d == 0:
            out.append(ch)
    return ''.join(out)


def extract(src_path, wanted):
    text = open(src_path, encoding='utf-8', errors='replace').read()
    # neutralise `#ifdef __cplusplus / extern "C" { / #endif` guards (and their closers),
    # which would otherwise leave the scanner one brace deep for the whole file
    def _blank(m):
        return re.sub(r'[^\n]', ' ', m.group(0))
    text = re.sub(r'#\s*ifdef\s+__cplusplus\s*\n\s*extern\s+"C"\s*\{[^\n]*\n\s*#\s*endif[^\n]*', _blank, text)
    text = re.sub(r'#\s*ifdef\s+__cplusplus\s*\n\s*\}[^\n]*\n\s*#\s*endif[^\n]*', _blank, text)
    chunks = _scan_chunks(text)
    found = {}
    out = []
    for (a, b, kind) in chunks:
        body = text[a:b]
        for nm in _chunk_names(body, kind):
            if nm in wanted and nm not in found:
                found[nm] = True
                # include a directly preceding OpenMP declare-simd / clone macro line
                out.append((a, body))
                break
    missing = [w for w in wanted if w not in found]
    return out, missing


def main():
    if len(sys.argv) < 4:
        print(__doc__)
        sys.exit(2)
    src, dst = sys.argv[1], sys.argv[2]
    wanted = sys.argv[3:]
    items, missing = extract(src, wanted)
    if missing:
        sys.stderr.write("extract.py: %s: not found: %s\n" % (src, ' '.join(missing)))
        sys.exit(1)
    with open(dst, 'w') as f:
        f.write("/* generated by oracle/extract.py from %s -- scratch, never commit */\n" % src)
        for _, body in items:
            f.write(body)
            f.write("\n\n")


if __name__ == '__main__':
    main()
