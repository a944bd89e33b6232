"""Loader for the reference's golden CSV fixtures (copied as data into tests/golden/).

Format (reference lib/src/test/resources/cl100k_base_encodings.csv:1): header
`input,output,outputMaxTokens10`; list columns are quoted "[1, 2, 3]" strings parsed the way
reference/TestUtils.java:9-16 does.  The last two rows use ", " separators, hence skipinitialspace.
"""
import csv
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ENCODING_NAMES = ["cl100k_base", "r50k_base", "p50k_base", "p50k_edit"]


def parse_encoding_string(s):
    s = s.strip()[1:-1].replace(" ", "")
    return [int(x) for x in s.split(",")] if s else []


def load_rows(name):
    path = os.path.join(GOLDEN, name + "_encodings.csv")
    rows = []
    with open(path, newline="", encoding="utf-8") as f:
        r = csv.reader(f, skipinitialspace=True)
        next(r)
        for row in r:
            if not row:
                continue
            rows.append((row[0], parse_encoding_string(row[1]), parse_encoding_string(row[2])))
    return rows
