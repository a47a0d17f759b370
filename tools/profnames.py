"""Kernel names as rocprofv3 prints them -> short, complete names for the committed summaries.

r03's summaries cut a name at its first '(' and so turned `dg::k_locate_topk<(unsigned int)1152>(...)` into `dg::k_locate_topk<`
(and several rows into a bare `dg::`): the parameter list is the LAST balanced parenthesis group, not the first."""
import re


def short_kernel_name(name: str, limit: int = 120) -> str:
    s = name.strip()
    s = re.sub(r"\s*\[clone [^\]]*\]$", "", s)
    s = re.sub(r"\.kd$", "", s)
    if s.startswith("void "):
        s = s[5:]
    if s.endswith(")"):
        depth = 0
        for i in range(len(s) - 1, -1, -1):
            if s[i] == ")":
                depth += 1
            elif s[i] == "(":
                depth -= 1
                if depth == 0:
                    s = s[:i]
                    break
    s = re.sub(r"\((?:unsigned int|int|unsigned long|bool|unsigned char)\)", "", s)  # `<(unsigned int)1152>` -> `<1152>`
    if "rocprim" in s:
        s = "rocprim::" + re.sub(r".*detail::", "", re.sub(r"<.*", "", s))
    s = re.sub(r"at::native::.*", "at::native::(torch kernel of the synthetic input generator)", s)
    return s[:limit]


if __name__ == "__main__":
    for t in ["void dg::k_locate_topk<(unsigned int)1152>(dg::FmView, dg::BigJob const*, unsigned int const*) [clone .kd]",
              "void dg::k_verify_memo<7, 8>(dg::FmView, dg::Batch, dg::VerifyArgs, dg::Counters*, unsigned int)",
              "dg::k_prepare(dg::Batch, unsigned int*, unsigned int*, unsigned int*, unsigned int*)",
              "void dg::k_search1s<true>(dg::FmView, dg::Batch, dg::SearchOut, dg::FlatSel, unsigned int, unsigned int, unsigned int, unsigned int)",
              "void dg::k_x<(bool)1, (int)2>(void (*)(int), int)"]:
        print(short_kernel_name(t))
