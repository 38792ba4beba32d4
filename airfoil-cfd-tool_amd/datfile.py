"""Airfoil ``.dat`` reader — the "`.dat` coords" half of the page's input surface.

Behavioural twin of the reference back end's parser, ``main.py:59-180`` of
583phoenix-hue/Airfoil-CFD-Tool (``parse_dat_file`` 59-113, ``detect_and_merge_sections``
116-180) and of the upload limits around it (``main.py:39-45, 551-582``): same function
names, same return shape ``(coords, fixes)``, same repair messages, same error behaviour
(the reference raises ``fastapi.HTTPException(400, detail)``; here :class:`DatParseError`
carries the same ``status_code``/``detail``).  Pure host code, no kernel.  Pinned by
tests/golden/datfile_cases.json, produced by running the reference's own two functions
(AST-extracted by oracle/make_goldens.py) on the same input texts.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

# main.py:39-41
MAX_FILE_SIZE = 1 * 1024 * 1024
MAX_POINTS = 500
MIN_POINTS = 10

# main.py:85: window a coordinate must fall in to count as airfoil data
_X_RANGE = (-0.5, 1.5)
_Y_RANGE = (-1.0, 1.0)

Point = List[float]


class DatParseError(ValueError):
    """What the reference signals with HTTPException(status_code=400, detail=...)."""

    def __init__(self, detail: str, status_code: int = 400):
        super().__init__(detail)
        self.detail = detail
        self.status_code = status_code


def _scan_lines(lines: Sequence[str]):
    """main.py:73-92: classify every line as coordinate / junk / out of range."""
    points: List[Point] = []
    junk = 0
    outside = 0
    for line in lines:
        tokens = line.split()
        if not tokens:
            continue                      # blank lines do not count as skipped
        if len(tokens) < 2:
            junk += 1
            continue
        try:
            x, y = float(tokens[0]), float(tokens[1])
        except ValueError:
            junk += 1
            continue
        if _X_RANGE[0] <= x <= _X_RANGE[1] and _Y_RANGE[0] <= y <= _Y_RANGE[1]:
            points.append([x, y])
        else:
            outside += 1
    return points, junk, outside


def detect_and_merge_sections(data_lines: Sequence[Point]) -> Tuple[List[Point], List[str]]:
    """main.py:116-180: Lednicer (two LE->TE sections) -> one Selig loop; reversed
    winding -> TE->upper->LE->lower->TE; a closed trailing edge is kept."""
    pts = list(data_lines)
    xs = [p[0] for p in pts]
    fixes: List[str] = []

    # a jump from the trailing edge (x > 0.5) straight back to the nose (x < 0.01) = second section
    split = next((i for i in range(1, len(pts)) if xs[i] < 0.01 and xs[i - 1] > 0.5), None)
    if split is not None:
        upper, lower = pts[:split], pts[split:]
        fixes.append(
            f"Lednicer format detected and converted: two-section format "
            f"({len(upper)} upper + {len(lower)} lower points) merged into "
            f"a single Selig-format loop for XFOIL")
        if upper[0][0] <= upper[-1][0]:           # stored LE->TE: walk it TE->LE
            upper = upper[::-1]
        if lower[0][0] > lower[-1][0]:            # stored TE->LE: walk it LE->TE
            lower = lower[::-1]
        if lower and abs(lower[0][0]) < 0.001 and abs(lower[0][1]) < 0.001:
            lower = lower[1:]
            fixes.append("Duplicate leading-edge point removed from Lednicer lower section")
        return upper + lower, fixes

    if xs[0] > 0.99 and xs[-1] > 0.99:            # TE ... TE: one loop; check its winding
        nose = xs.index(min(xs))
        if nose > 0 and not pts[nose - 1][1] > 0:
            fixes.append(
                "Winding order corrected: coordinates were in reversed order "
                "(TE→lower→LE→upper→TE) and have been reversed to the correct "
                "Selig order (TE→upper→LE→lower→TE)")
            return pts[::-1], fixes
    return pts, fixes


def parse_dat_lines(lines: Sequence[str]) -> Tuple[List[Point], List[str]]:
    points, junk, outside = _scan_lines(lines)
    fixes: List[str] = []
    if junk:
        fixes.append(f"Non-coordinate lines skipped: {junk} header/comment line(s) removed")
    if outside:
        fixes.append(f"Out-of-range points filtered: {outside} point(s) outside valid bounds removed")
    if len(points) < MIN_POINTS:
        raise DatParseError(f"Insufficient valid coordinates. Found {len(points)} points.")
    coords, geom_fixes = detect_and_merge_sections(points)
    fixes += geom_fixes
    if not fixes:
        fixes = ["No changes made — file was already in valid Selig format"]
    return coords, fixes


def parse_dat_file(file_path: str) -> Tuple[List[Point], List[str]]:
    """main.py:59-113.  Any failure (unreadable file included) is a 400-class DatParseError."""
    try:
        with open(file_path, "r") as fh:
            lines = fh.readlines()
        return parse_dat_lines(lines)
    except DatParseError:
        raise
    except Exception as exc:                       # main.py:110-113
        raise DatParseError(f"Failed to parse file: {exc}") from exc


def load_dat(file_path: str) -> Tuple[List[Point], List[str]]:
    """parse_dat_file behind the upload checks of main.py:556-582: extension, size, point count.
    The result is the `coords_after` the reference hands to build_lbm_component (main.py:607-608)."""
    if not str(file_path).endswith(".dat"):
        raise DatParseError("Only .dat files accepted")
    if os.path.getsize(file_path) > MAX_FILE_SIZE:
        raise DatParseError(f"File too large (max {MAX_FILE_SIZE / (1024 * 1024)}MB)")
    coords, fixes = parse_dat_file(file_path)
    if len(coords) > MAX_POINTS:
        raise DatParseError(f"Too many points (max {MAX_POINTS})")
    return coords, fixes
