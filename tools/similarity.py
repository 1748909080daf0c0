#!/usr/bin/env python3
"""Share of a product routine's statements that also occur, in order, in the reference routine of the same name
(comments and blank lines stripped, whitespace-insensitive, continuation lines joined).  Study aid: the reference is
only read.   python tools/similarity.py [/root/reference/diaglib.f90]"""
import difflib, re, sys
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/diaglib.f90"
MINE = "diaglib_amd/fortran/diaglib.f90"


def routines(path):
    txt = open(path).read().splitlines()
    out, cur, name = {}, None, None
    for ln in txt:
        code = ln.split("!")[0] if not re.match(r"\s*!\$", ln) else ln
        m = re.match(r"\s*(?:recursive\s+)?(subroutine|function)\s+(\w+)", code, re.I)
        if m and cur is None:
            name, cur = m.group(2).lower(), []
        if cur is not None:
            cur.append(code)
            if re.match(r"\s*end\s+(subroutine|function)", code, re.I):
                out[name] = cur; cur = None
    return out


def norm(lines):
    joined, buf = [], ""
    for ln in lines:
        s = re.sub(r"\s+", "", ln).lower()
        if not s:
            continue
        if s.startswith("&"):
            s = s[1:]
        buf += s
        if buf.endswith("&"):
            buf = buf[:-1]; continue
        joined.append(buf); buf = ""
    return joined


ref, mine = routines(REF), routines(MINE)
pairs = [("davidson_driver", ["davidson_driver", "davidson_core"]), ("gen_david_driver", ["gen_david_driver", "davidson_core"]),
         ("lobpcg_driver", ["lobpcg_driver"]), ("caslr_driver", ["caslr_driver", "lr_core"]),
         ("caslr_eff_driver", ["caslr_eff_driver", "lr_core"]), ("ortho_cd", ["ortho_cd"]), ("ortho_vs_x", ["ortho_vs_x"]),
         ("b_ortho", ["b_ortho"]), ("b_ortho_vs_x", ["b_ortho_vs_x"]), ("ortho", ["ortho"])]
for rname, mnames in pairs:
    r = norm(ref[rname])
    for mn in mnames:
        m = norm(mine[mn])
        sm = difflib.SequenceMatcher(None, m, r, autojunk=False)
        same = sum(b.size for b in sm.get_matching_blocks())
        print(f"{mn:18s} vs reference {rname:18s}: {same:4d} of {len(m):4d} statements identical = {100.0 * same / max(1, len(m)):5.1f} %")
