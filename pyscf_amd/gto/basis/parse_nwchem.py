"""NWChem-format basis-set text parser.

Produces the same internal format as the reference's parser
(pyscf/gto/basis/parse_nwchem.py:105-155): a list of shells
``[l, [exp, c1, c2, ...], ...]`` sorted by angular momentum, "SP" blocks split
into an s and a p shell, zero-coefficient primitives dropped
(``remove_zero``, parse_nwchem.py:288-303).  Contractions are NOT re-optimised
(OPTIMIZE_CONTRACTION=False, pyscf/gto/basis/__init__.py:39).
"""
import re

MAXL = 15
SPDF = 'SPDFGHIKLMNORTU'
MAPSPDF = {c: l for l, c in enumerate(SPDF)}
_DELIM = re.compile(r'# *BASIS SET.*\n|END\n')


def _std_symbol(symb):
    symb = ''.join(c for c in symb if c.isalpha())
    return symb[0].upper() + symb[1:].lower()


def search_seg(text, symb):
    """Lines of the block that belongs to element `symb` (parse_nwchem.py:156-183)."""
    symb = _std_symbol(symb)
    for dat in re.split(_DELIM, text):
        head = dat.split(None, 1)
        if not head:
            continue
        if head[0] == symb:
            return [x for x in dat.splitlines() if x and 'END' not in x]
        if head[0][0] == '#':
            lines = dat.splitlines()
            for i, line in enumerate(lines):
                if not line or line.lstrip()[0] == '#':
                    continue
                if line.split(None, 1)[0] == symb:
                    return [x for x in lines[i:] if x and 'END' not in x]
                break
    return []


def parse_lines(lines):
    shells = [[] for _ in range(MAXL)]
    key = None
    cur = None
    for line in lines:
        dat = line.strip()
        if not dat or dat.startswith('#'):
            continue
        if dat[0].isalpha():
            keys = dat.split()
            key = (keys[0] if len(keys) == 1 else keys[1]).upper()
            if key == 'SP':
                shells[0].append([0])
                shells[1].append([1])
            elif key in MAPSPDF:
                cur = [MAPSPDF[key]]
                shells[MAPSPDF[key]].append(cur)
            else:
                raise ValueError('Not basis data: %s' % line)
        else:
            vals = [float(x) for x in dat.replace('D', 'e').replace('d', 'e').split()]
            if key is None:
                raise ValueError('Not basis data')
            if key == 'SP':
                shells[0][-1].append([vals[0], vals[1]])
                shells[1][-1].append([vals[0], vals[2]])
            else:
                cur.append(vals)
    out = []
    for bs in shells:
        for b in bs:
            ec = [e_c for e_c in b[1:] if any(c != 0 for c in e_c[1:])]
            if ec:
                out.append([b[0]] + ec)
    if not out:
        raise KeyError('Basis data not found')
    return out


def parse(text, symb=None):
    """Parse an NWChem-format string; if `symb` is given select that element."""
    if symb is not None:
        lines = search_seg(text, symb)
        if not lines:
            raise KeyError('Basis not found for %s' % symb)
    else:
        lines = [x for x in text.splitlines() if x and 'END' not in x and 'BASIS' not in x]
    return parse_lines(lines)
