"""Print the instruction mix of every basic block that holds MFMAs in one kernel of a hipcc -S listing.

usage: isa_blocks.py listing.s mangled_kernel_name
letters: M mfma, r ds_read, w ds_write, G global/buffer load, S global store, W s_waitcnt, B s_barrier,
         v other vector ALU, s scalar, x scratch
"""
import re
import sys


def main(path, kernel):
    text = open(path).read().split("\n")
    start = next(i for i, l in enumerate(text) if l.startswith(kernel + ":"))
    end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
    blocks, cur = [], ["entry", []]
    for l in text[start:end]:
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append(cur)
            cur = [l.split(":")[0], []]
        elif l.startswith("\t") and not l.strip().startswith(";") and not l.strip().startswith("."):
            cur[1].append(l.strip())
    blocks.append(cur)

    def letter(i):
        if i.startswith("v_mfma"): return "M"
        if i.startswith("ds_read"): return "r"
        if i.startswith("ds_write"): return "w"
        if i.startswith("global_load") or i.startswith("buffer_load"): return "G"
        if i.startswith("global_store") or i.startswith("global_atomic"): return "S"
        if i.startswith("scratch_"): return "x"
        if i.startswith("s_waitcnt"): return "W"
        if i.startswith("s_barrier"): return "B"
        if i.startswith("v_"): return "v"
        return "s"

    for name, ins in blocks:
        mf = sum(1 for i in ins if i.startswith("v_mfma"))
        if mf:
            print(name, "n=%d mfma=%d" % (len(ins), mf))
            print("  " + "".join(letter(i) for i in ins))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
