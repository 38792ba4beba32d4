#!/usr/bin/env python3
"""tests/golden/datfile_cases.json: the REFERENCE's own parse_dat_file / detect_and_merge_sections
(main.py:59-180), AST-extracted from /root/reference/main.py and executed here, applied to input
texts authored in this file.  TEST INFRASTRUCTURE; runs only in the build container (main.py itself
cannot be imported: slowapi / python-multipart are absent)."""
import ast
import json
import logging
import math
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/main.py"


def reference_functions():
    from fastapi import HTTPException
    tree = ast.parse(open(REF).read())
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("parse_dat_file", "detect_and_merge_sections")]
    assert len(wanted) == 2
    ns = {"HTTPException": HTTPException, "logger": logging.getLogger("ref")}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), REF, "exec"), ns)
    return ns["parse_dat_file"], ns["detect_and_merge_sections"], HTTPException


def naca(m, p, t, n, te=-0.1015):
    up, lo = [], []
    for i in range(n + 1):
        x = 0.5 * (1 - math.cos(math.pi * i / n))
        yt = 5 * t * (0.2969 * math.sqrt(x) - 0.126 * x - 0.3516 * x * x + 0.2843 * x ** 3 + te * x ** 4)
        yc = 0.0
        if m > 0:
            yc = m / p ** 2 * (2 * p * x - x * x) if x < p else m / (1 - p) ** 2 * ((1 - 2 * p) + 2 * p * x - x * x)
        up.append((x, yc + yt))
        lo.append((x, yc - yt))
    return up, lo


def fmt(pts, sep="  ", prec=6):
    return [f"{x:.{prec}f}{sep}{y:.{prec}f}" for x, y in pts]


def cases():
    up, lo = naca(0.02, 0.4, 0.12, 20)
    selig = up[::-1] + lo[1:]
    c = {}
    c["selig_with_header"] = ["NACA 2412 (authored)"] + fmt(selig)
    c["selig_no_header"] = fmt(selig)
    c["selig_blank_and_comment_lines"] = ["# airfoil", "", "   "] + fmt(selig[:10]) + ["", "; mid comment"] + fmt(selig[10:]) + [""]
    c["selig_tab_separated"] = ["NACA"] + fmt(selig, sep="\t", prec=4)
    c["selig_reversed_winding"] = ["reversed"] + fmt(selig[::-1])
    c["selig_closed_te_loop"] = ["closed"] + fmt([(1.0, 0.0)] + selig[1:-1] + [(1.0, 0.0)])
    c["lednicer_counts_header"] = ["LEDNICER", "      21.       21.", ""] + fmt(up) + [""] + fmt(lo)
    c["lednicer_upper_descending"] = ["LEDNICER rev upper"] + fmt(up[::-1]) + fmt(lo)
    c["lednicer_lower_descending"] = ["LEDNICER rev lower"] + fmt(up) + fmt([(0.0, 0.0)] + lo[::-1][:-1])
    c["lednicer_no_duplicate_le"] = ["LEDNICER"] + fmt(up) + fmt([(0.005, -0.009)] + lo[2:])
    c["out_of_range_points_filtered"] = ["junk"] + fmt(selig[:20]) + ["2.5  0.1", "-0.7 0.0", "0.5 1.5"] + fmt(selig[20:])
    c["extra_columns_ignored"] = [f"{x:.5f} {y:.5f} 0.0 extra" for x, y in selig]
    c["scientific_notation"] = [f"{x:.6e} {y:.6e}" for x, y in selig]
    c["le_first_single_section"] = fmt(lo[::-1][:-1][::-1] + [])       # LE->TE only, never TE..TE
    c["too_few_points"] = ["short"] + fmt(selig[:6])
    c["all_out_of_range"] = ["2.0  0.5", "3.0  0.1", "-2.0  0.0"]
    c["single_token_lines"] = ["1.0", "0.5", "abc"] + fmt(selig)
    c["empty_file"] = []
    c["nan_and_inf_tokens"] = ["nan 0.1", "inf 0.0"] + fmt(selig)
    c["te_to_te_nose_first_index_zero"] = fmt([(0.995, 0.001), (0.5, 0.05), (0.1, 0.03), (0.2, -0.02), (0.5, -0.03), (0.8, -0.01),
                                               (0.9, -0.005), (0.95, -0.002), (0.97, -0.001), (0.98, -0.0005), (0.995, -0.001)])
    return c


def main():
    parse, merge, HTTPException = reference_functions()
    out = {"cases": [], "merge_cases": []}
    for name, lines in cases().items():
        text = "\n".join(lines)
        with tempfile.NamedTemporaryFile("w", suffix=".dat", delete=False) as fh:
            fh.write(text)
        rec = {"name": name, "text": text}
        try:
            coords, fixes = parse(fh.name)
            rec.update(coords=[[float(x), float(y)] for x, y in coords], fixes=fixes)
        except HTTPException as e:
            rec.update(error={"status_code": e.status_code, "detail": e.detail})
        os.unlink(fh.name)
        out["cases"].append(rec)
    try:
        parse("/nonexistent/path/file.dat")
    except HTTPException as e:
        out["missing_file"] = {"status_code": e.status_code, "detail_prefix": e.detail.split(":")[0]}
    # direct detect_and_merge_sections inputs (shapes of test_main.py:112-199, coordinates authored here)
    up = [[0.0, 0.0], [0.3, 0.05], [0.6, 0.04], [0.9, 0.012], [1.0, 0.002]]
    lo = [[0.0, 0.0], [0.3, -0.03], [0.6, -0.025], [0.9, -0.008], [1.0, -0.002]]
    loop = [[1.0, 0.002], [0.6, 0.04], [0.3, 0.05], [0.0, 0.0], [0.3, -0.03], [0.6, -0.025], [1.0, -0.002]]
    for name, data in (("lednicer", up + lo), ("selig", loop), ("reversed", loop[::-1]),
                       ("closed", [[1.0, 0.0]] + loop[1:-1] + [[1.0, 0.0]]), ("open_ended", loop[:-2])):
        merged, fixes = merge([list(p) for p in data])
        out["merge_cases"].append({"name": name, "data": data, "merged": merged, "fixes": fixes})
    path = os.path.join(ROOT, "tests", "golden", "datfile_cases.json")
    json.dump(out, open(path, "w"), indent=1, ensure_ascii=False)
    print(f"{path}: {len(out['cases'])} file cases, {len(out['merge_cases'])} merge cases;",
          sum(1 for c in out["cases"] if "error" in c), "error cases")


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("reference not mounted")
    main()
