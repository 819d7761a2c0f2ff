"""Parses the one known-answer vector the reference itself holds into tests/golden/docs_kat.npz (data only).

    python oracle/gen_golden_docs.py            (needs /root/reference; run in the build container)

Inputs, all text dumps written by the reference's src/utils/debug.py from one 20x20 map, depth 500:
  docs/map.txt          print_rails  (debug.py:4-34):   `[n]` = switch n, bare number = rail id, `.` = empty
  docs/cell_seq.txt     print_cells_sequence (:63-85):  per agent the list of predicted cells, one per time step
  docs/z-bitmaps-0.txt  print_bitmaps (:36-61):         per agent and rail one row of depth+1 columns, [+] [-] or blank
Agent 0's bitmap rows carry hand-written annotations in the reference's file; agents 1-3 are clean.

Output arrays: node_id [H,W] int32 (-1 = no switch), cell_rail [H,W] int32 (-1 = no rail), cell_seq [4,T1,2] int32,
bitmaps [4,R,T1] int8 (agent 0: rows parsed best-effort, do not use), clean_agents = [1,2,3].
"""
import os
import re
import sys

import numpy as np

REF = "/root/reference/docs"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "docs_kat.npz")


def parse_map(path):
    rows = [l for l in open(path, encoding="utf-8").read().replace("\r", "").split("\n") if l.strip()]
    cells = [re.findall(r"\[\d+\]|\d+|·", l) for l in rows]
    H, W = len(cells), len(cells[0])
    assert all(len(c) == W for c in cells)
    node_id = -np.ones((H, W), np.int32)
    cell_rail = -np.ones((H, W), np.int32)
    for r in range(H):
        for c in range(W):
            tok = cells[r][c]
            if tok.startswith("["):
                node_id[r, c] = int(tok[1:-1])
            elif tok != "·":
                cell_rail[r, c] = int(tok)
    return node_id, cell_rail


def parse_cell_seq(path):
    text = open(path, encoding="utf-8").read()
    seqs = []
    for m in re.finditer(r"CELLSEQ: \[(.*?)\] \n", text):
        seqs.append(np.array([(int(a), int(b)) for a, b in re.findall(r"\((\d+), (\d+)\)", m.group(1))], np.int32))
    return np.stack(seqs)


def parse_bitmaps(path, T1):
    lines = open(path, encoding="utf-8").read().split("\n")
    agents, cur = [], None
    for l in lines:
        if l.startswith("AGENT"):
            cur = []
            agents.append(cur)
        elif l.startswith(">"):
            body = l.split(":", 1)[1][2:]                      # '>{:^3}:  ' then T1 fields of 3 characters
            row = np.zeros(T1, np.int8)
            for t in range(T1):
                f = body[3 * t:3 * t + 3]
                row[t] = 1 if f == "[+]" else -1 if f == "[-]" else 0
            cur.append(row)
    return np.stack([np.stack(a) for a in agents])


def main():
    node_id, cell_rail = parse_map(os.path.join(REF, "map.txt"))
    seq = parse_cell_seq(os.path.join(REF, "cell_seq.txt"))
    bm = parse_bitmaps(os.path.join(REF, "z-bitmaps-0.txt"), seq.shape[1])
    print("map", node_id.shape, "switches", int(node_id.max()) + 1, "rails on the map", len(np.unique(cell_rail)) - 1,
          "cell_seq", seq.shape, "bitmaps", bm.shape)
    np.savez_compressed(OUT, node_id=node_id, cell_rail=cell_rail, cell_seq=seq, bitmaps=bm,
                        clean_agents=np.array([1, 2, 3], np.int32))
    print("wrote", OUT)


if __name__ == "__main__":
    sys.exit(main())
