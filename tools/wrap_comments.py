#!/usr/bin/env python3
"""Readability pass over the kernel sources: a trailing `// comment` that makes a line longer than LIMIT moves to its own line(s) above the statement,
comment-only lines are re-wrapped at LIMIT columns, and code lines that are still longer are broken where white space means nothing: behind a `;`
outside parentheses, else behind a `,` inside them (argument lists).  Lines inside macros (ending in a backslash), lines with an odd number of quotes
and preprocessor lines are left alone.  usage: wrap_comments.py file... [--limit=N]"""
import re
import sys
import textwrap

LIMIT = 170


def wrap(indent, text):
    out = []
    for ln in textwrap.wrap(text, max(40, LIMIT - len(indent) - 3), break_long_words=False, break_on_hyphens=False):
        out.append(indent + "// " + ln)
    return out or [indent + "//"]


def split_comment(line):
    """index of the `//` that starts a trailing comment, or -1"""
    in_s = False; q = ""; i = 0
    while i < len(line) - 1:
        c = line[i]
        if in_s:
            if c == "\\": i += 2; continue
            if c == q: in_s = False
        elif c in "\"'": in_s = True; q = c
        elif c == "/" and line[i + 1] == "/": return i
        i += 1
    return -1


def break_code(line, indent):
    """pieces of a code line (with its trailing comment, if any, on the last piece), each at most LIMIT columns where a break point allows"""
    ci = split_comment(line)
    code = line if ci < 0 else line[:ci]
    tail = "" if ci < 0 else line[ci:]
    pieces = []; cont = indent + "        "
    cur = code.rstrip()
    first = True
    while len(cur) > LIMIT:
        depth = 0; in_s = False; q = ""; best_semi = -1; best_comma = -1; i = 0
        lim = LIMIT
        while i < len(cur) and i < lim:
            c = cur[i]
            if in_s:
                if c == "\\": i += 2; continue
                if c == q: in_s = False
            elif c in "\"'": in_s = True; q = c
            elif c in "([{" and c != "{": depth += 1
            elif c in ")]" : depth -= 1
            elif c == ";" and depth == 0 and i + 1 < len(cur) and cur[i + 1] == " ": best_semi = i + 1
            elif c == "," and depth >= 1 and i + 1 < len(cur) and cur[i + 1] == " ": best_comma = i + 1
            i += 1
        cut = best_semi if best_semi > len(indent) + 40 else best_comma
        if cut <= len(indent) + 20 or in_s: break
        pieces.append(cur[:cut].rstrip()); cur = (cont if True else indent) + cur[cut:].lstrip(); first = False
    pieces.append(cur + ((" " + tail.strip()) if tail.strip() else ""))
    return pieces


def process(path):
    src = open(path).read().split("\n"); out = []; changed = 0
    for k, line in enumerate(src):
        if len(line) <= LIMIT or line.rstrip().endswith("\\") or line.lstrip().startswith("#") or (k and src[k - 1].rstrip().endswith("\\")):
            out.append(line); continue
        i = split_comment(line)
        indent = re.match(r"\s*", line).group(0)
        if i < 0:
            if line.count('"') % 2 == 0:
                pcs = break_code(line, indent); out.extend(pcs); changed += len(pcs) > 1
            else:
                out.append(line)
            continue
        code, com = line[:i].rstrip(), line[i + 2:].strip()
        if not code:                                            # a comment-only line: re-wrap (keep list / continuation indentation inside the comment)
            inner = re.match(r"\s*", line[i + 2:]).group(0)
            body = line[i + 2:].strip()
            lines = textwrap.wrap(body, max(40, LIMIT - len(indent) - 2 - len(inner)), break_long_words=False, break_on_hyphens=False)
            out.extend(indent + "//" + inner + ln for ln in lines); changed += 1
        elif '"' in code and code.count('"') % 2:               # (an odd number of quotes: not sure where the string ends)
            out.append(line)
        else:
            out.extend(wrap(indent, com)); out.extend(break_code(code, indent)); changed += 1
    open(path, "w").write("\n".join(out))
    return changed


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    for a in sys.argv[1:]:
        if a.startswith("--limit"): LIMIT = int(a.split("=")[1])
    for p in args:
        print(p, process(p))
