#!/usr/bin/env python3
"""One-off source transformation (kept for the record / re-runs): wrap every C-ABI entry point of csrc/*.hip in a
function-try-block so that no C++ exception (std::bad_alloc from std::vector / std::string, std::system_error from
std::thread) can unwind through the `extern "C"` boundary into a host built with panic = "abort" (Cargo.toml:64).

    int ab_foo(ab_ctx *ctx, ...) {          int ab_foo(ab_ctx *ctx, ...) try {
        ...                           ->        ...
    }                                       } AB_CATCH(ctx)

AB_CATCH (ab_common.hpp) maps bad_alloc -> AB_ERR_NOMEM and anything else -> AB_ERR_INVALID, with the message in
ab_last_error.  Functions without a context use AB_CATCH_NOCTX.  Idempotent: already wrapped functions are skipped."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "astroburst_hip.h")).read()
API = set(re.findall(r"AB_API[^;(]*?\b(ab_[a-z0-9_]+)\s*\(", HEADER))
SKIP = {"ab_ctx_destroy", "ab_comm_destroy", "ab_last_error", "ab_version", "ab_ctx_get_stream", "ab_comm_rank", "ab_comm_size",
        "ab_comm_collectives_issued", "ab_subframe_weight_config_default", "ab_batch_stack_config_default",
        "ab_normalize_subframe_weights"}   # void / pointer / trivially non-throwing getters


def match_brace(s, i):
    """index of the brace closing the one at s[i] (skips strings, chars and comments)"""
    depth, n = 0, len(s)
    while i < n:
        c = s[i]
        if s.startswith("//", i):
            i = s.index("\n", i)
            continue
        if s.startswith("/*", i):
            i = s.index("*/", i) + 2
            continue
        if c == '"' or c == "'":
            j = i + 1
            while s[j] != c:
                j += 2 if s[j] == "\\" else 1
            i = j + 1
            continue
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return i
        i += 1
    raise ValueError("unbalanced")


def main():
    changed = 0
    for path in sorted(glob.glob(os.path.join(ROOT, "astroburst_amd", "csrc", "*.hip"))):
        s = open(path).read()
        out, pos = [], 0
        for m in re.finditer(r'^(?:extern "C" )?(?:int|uint64_t)\s+(ab_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*\{', s, re.M):
            name = m.group(1)
            if name not in API or name in SKIP or m.start() < pos:
                continue
            open_i = m.end() - 1
            close_i = match_brace(s, open_i)
            has_ctx = re.search(r"\bab_ctx\s*\*\s*ctx\b", m.group(2)) is not None
            out.append(s[pos:open_i] + "try {" + s[open_i + 1:close_i] + "} " + ("AB_CATCH(ctx)" if has_ctx else "AB_CATCH_NOCTX"))
            pos = close_i + 1
            changed += 1
        out.append(s[pos:])
        new = "".join(out)
        if new != s:
            open(path, "w").write(new)
    print(f"wrapped {changed} entry points")


if __name__ == "__main__":
    main()
