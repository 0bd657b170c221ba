"""Reader for the reference's flat configuration files (`.knowledge`, `.settings`, `.platform`).

Reference: `key = value` lines, `//` comments, `import '<file>'`, strings in single or double quotes, booleans, integers,
reals (Compiler/src/exastencils/parsers/config/Settings_Parser.scala; usage Compiler/src/Main.scala:41-50), followed by
the constraint pass Knowledge.update() (Compiler/src/exastencils/config/Knowledge.scala:866-1078), of which the few
derivations the hot path needs are reproduced in `derive()`: totals of blocks/fragments, `domain_fragmentLength_*`
defaults, the level range.  The flag names are the reference's, so its own files drive this package.
"""
from __future__ import annotations

import os
import re
from typing import Dict, Optional

_LINE = re.compile(r"^\s*([A-Za-z_][A-Za-z0-9_]*)\s*=\s*(.+?)\s*$")
_IMPORT = re.compile(r"^\s*import\s+['\"](.+?)['\"]\s*$")


def _value(tok: str):
    tok = tok.strip()
    if (tok[0] == tok[-1]) and tok[0] in "'\"":
        return tok[1:-1]
    if tok in ("true", "false"):
        return tok == "true"
    try:
        return int(tok)
    except ValueError:
        pass
    try:
        return float(tok)
    except ValueError:
        return tok


def parse_text(text: str, base_dir: Optional[str] = None, into: Optional[Dict] = None) -> Dict:
    out = {} if into is None else into
    text = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)     # block comments (Settings_Parser.scala)
    for raw in text.splitlines():
        line = raw.split("//", 1)[0].strip()
        if not line:
            continue
        m = _IMPORT.match(line)
        if m:
            if base_dir is None:
                raise ValueError("import %r needs a base directory" % m.group(1))
            parse_file(os.path.join(base_dir, m.group(1)), out)
            continue
        m = _LINE.match(line)
        if not m:
            raise ValueError("cannot parse configuration line: %r" % raw)
        out[m.group(1)] = _value(m.group(2))
    return out


def parse_file(path: str, into: Optional[Dict] = None) -> Dict:
    with open(path) as f:
        return parse_text(f.read(), os.path.dirname(os.path.abspath(path)), into)


def derive(k: Dict) -> Dict:
    """The derived quantities the hot path reads (defaults as in Knowledge.scala:35-132)."""
    nd = int(k.get("dimensionality", 3))
    ax = "xyz"
    nb = tuple(int(k.get("domain_rect_numBlocks_" + ax[d], 1)) if d < nd else 1 for d in range(3))
    nf = tuple(int(k.get("domain_rect_numFragsPerBlock_" + ax[d], 1)) if d < nd else 1 for d in range(3))
    fl = tuple(int(k.get("domain_fragmentLength_" + ax[d], 1)) if d < nd else 1 for d in range(3))
    return {
        "dimensionality": nd,
        "min_level": int(k.get("minLevel", 0)),
        "max_level": int(k.get("maxLevel", 0)),
        "num_blocks": nb,
        "frags_per_block": nf,
        "frag_len": fl,
        "frags_total": tuple(nb[d] * nf[d] for d in range(3)),
        "cells_per_dim_finest": tuple(nb[d] * nf[d] * fl[d] * (1 << int(k.get("maxLevel", 0))) if d < nd else 0 for d in range(3)),
        "mpi_ranks": int(k.get("mpi_numThreads", nb[0] * nb[1] * nb[2])),
        "omp_threads": int(k.get("omp_numThreads", 1)),
        "cuda_block": tuple(int(k.get("cuda_blockSize_" + ax[d], (32, 4, 4)[d])) for d in range(3)),
        "comm_axis_neighbors_only": bool(k.get("comm_onlyAxisNeighbors", True)),
        "periodic": tuple(bool(k.get("domain_rect_periodic_" + ax[d], False)) if d < nd else False for d in range(3)),
    }


def domain_for_rank(k: Dict, rank: int, one_fragment_per_block: bool = True):
    """RectDomain of this process for a knowledge set: the reference's blocks, with each block's fragments merged into
    one fragment per GPU (fragLen * fragsPerBlock cells), the single-fragment-per-process mode of this package."""
    from .domain import RectDomain

    d = derive(k)
    if not one_fragment_per_block:
        raise NotImplementedError("several fragments per process")
    flen = tuple(d["frag_len"][i] * d["frags_per_block"][i] for i in range(3))
    return RectDomain(d["dimensionality"], d["num_blocks"], rank, flen)
