"""Static instruction mix and register / LDS use of kernels in a hipcc -save-temps gfx950 assembly file.
    python tools/isa_stats.py <file.s> <kernel-name-substring> [...]"""
import re
import sys
from collections import Counter


def main():
    s = open(sys.argv[1]).read()
    for name in sys.argv[2:]:
        for m in re.finditer(r"^(_Z\w*%s\w*):.*?\n(.*?)\.Lfunc_end" % re.escape(name), s, re.S | re.M):
            ins = [ln.strip() for ln in m.group(2).split("\n")]
            ins = [i for i in ins if i and not i.startswith((";", ".", "/")) and not i.split(";")[0].strip().endswith(":")]
            c = Counter(i.split()[0] for i in ins)
            print(m.group(1)[:90])
            print("  static instructions", len(ins))
            print("  ", c.most_common(30))
            k = re.search(r"\.amdhsa_kernel %s.*?\.end_amdhsa_kernel" % re.escape(m.group(1)), s, re.S)
            if k:
                for key in ("next_free_vgpr", "next_free_sgpr", "group_segment_fixed_size", "private_segment_fixed_size", "accum_offset"):
                    mm = re.search(r"\.amdhsa_%s (\d+)" % key, k.group(0))
                    print("  ", key, mm.group(1) if mm else None)


if __name__ == "__main__":
    main()
