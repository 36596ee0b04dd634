#!/usr/bin/env python3
"""Re-flow a Markdown file to a column limit so that it reads in a terminal and in a diff: paragraphs and list items are wrapped
(continuation lines indented under the item's text), tables one of whose rows exceeds the limit become lists (a table cell cannot
be wrapped: one item per row, the first cell as its title, the other cells as '<column>: <cell>' sub-items), code fences and short
tables are left alone.  usage: wrap_markdown.py <file> [limit=150]"""
import re
import sys
import textwrap


def wrap(text, first, rest, limit):
    w = textwrap.TextWrapper(width=limit, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)
    return w.fill(text).split("\n")


def cells(row):
    row = row.strip()
    if row.startswith("|"):
        row = row[1:]
    if row.endswith("|"):
        row = row[:-1]
    out, cur, tick, k = [], "", False, 0
    while k < len(row):
        ch = row[k]
        if ch == "\\" and k + 1 < len(row) and row[k + 1] == "|":      # an escaped bar is text (and needs no escape outside a table)
            cur += "|"; k += 2; continue
        if ch == "`":
            tick = not tick
        if ch == "|" and not tick:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
        k += 1
    out.append(cur.strip())
    return out


def main():
    path = sys.argv[1]
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    lines = open(path).read().split("\n")
    out, i, fence = [], 0, False
    while i < len(lines):
        l = lines[i]
        if l.lstrip().startswith("```"):
            fence = not fence
            out.append(l); i += 1; continue
        if fence:
            out.append(l); i += 1; continue
        if l.startswith("|"):
            j = i
            while j < len(lines) and lines[j].startswith("|"):
                j += 1
            block = lines[i:j]
            if max(len(b) for b in block) <= limit:
                out.extend(block)
            else:
                head = cells(block[0])
                for row in block[2:]:
                    c = cells(row)
                    out.extend(wrap("**" + c[0] + "**", "* ", "  ", limit))
                    for h, v in zip(head[1:], c[1:]):
                        if v:
                            out.extend(wrap("%s: %s" % (h, v), "  - ", "    ", limit))
                out.append("")
            i = j; continue
        if len(l) <= limit:
            out.append(l); i += 1; continue
        m = re.match(r"^(\s*)([*-]|\d+\.)\s+", l)
        if m:
            first = m.group(0)
            out.extend(wrap(l[len(first):], first, " " * len(first), limit))
        elif l.startswith("#"):
            out.append(l)
        else:
            ind = re.match(r"^\s*", l).group(0)
            out.extend(wrap(l.strip(), ind, ind, limit))
        i += 1
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
