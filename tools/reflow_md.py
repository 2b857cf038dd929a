#!/usr/bin/env python3
"""Reflow the prose of a markdown file to a column width (headings, tables, code fences and blank lines are kept as they are;
list items keep a hanging indent).  The text itself must not change: checked by comparing the whitespace-normalised contents.
usage: tools/reflow_md.py FILE [WIDTH=118]"""
import re, sys, textwrap

def reflow(text, width):
    out, para, fence = [], [], False
    def flush():
        if not para: return
        first = para[0]
        m = re.match(r'^(\s*)((?:[*+-]|\d+\.)\s+)?', first)
        lead = m.group(1) or ''
        bullet = m.group(2) or ''
        body = ' '.join([first[len(lead) + len(bullet):].strip()] + [l.strip() for l in para[1:]])
        # keep the two spaces this file puts after a full stop
        body = re.sub(r'\s+', ' ', body)
        body = re.sub(r'([.:]\)?) (?=[A-Z`*("\[])', r'\1  ', body) if '.  ' in ' '.join(para) else body
        hang = lead + ' ' * len(bullet) if bullet else lead
        # a continuation paragraph inside a list item arrives with its own indent in `lead`
        w = textwrap.TextWrapper(width=width, initial_indent=lead + bullet, subsequent_indent=hang,
                                 break_long_words=False, break_on_hyphens=False)
        out.extend(w.wrap(body))
        para.clear()
    for line in text.split('\n'):
        if line.startswith('```'):
            flush(); fence = not fence; out.append(line); continue
        if fence: out.append(line); continue
        if not line.strip() or line.startswith('#') or line.lstrip().startswith('|') or re.match(r'^\s*(---|===)', line):
            flush(); out.append(line); continue
        if re.match(r'^\s*(?:[*+-]|\d+\.)\s+', line): flush()
        para.append(line)
    flush()
    return '\n'.join(out)

if __name__ == '__main__':
    path = sys.argv[1]; width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
    src = open(path).read()
    dst = reflow(src, width)
    norm = lambda s: re.sub(r'\s+', ' ', s).strip()
    assert norm(src) == norm(dst), 'reflow changed the text'
    open(path, 'w').write(dst)
    print(path, len(src.split('\n')), '->', len(dst.split('\n')), 'lines')
