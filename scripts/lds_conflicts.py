"""Bank-conflict model for ds_read_b128 on gfx950 (64 banks x 4 B, four fixed 16-lane groups)."""
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]

def cycles_b128(addrs):
    """LDS cycles for one wave-instruction given 64 byte addresses (16 B each)."""
    tot = 0
    for g in GROUPS:
        banks = {}
        for l in g:
            a = addrs[l]
            for b in range(4):
                bank = ((a // 4) + b) % 64
                banks.setdefault(bank, set()).add(a // 4 + b)
        tot += max(len(v) for v in banks.values())
    return tot

def conv_x_read(stride, m, tapoff, ks, flat=True):
    addrs = []
    for l in range(64):
        l15, lq = l & 15, l >> 4
        if flat:
            px = min(m * 16 + l15, 195); y, x = divmod(px, 14)
        else:
            y, x = m, l15
        addrs.append((y * 16 + x + tapoff) * stride + lq * 16 + ks * 64)
    return addrs

if __name__ == '__main__':
    for stride in (272, 288, 304, 320, 336, 352):
        for flat in (True, False):
            tot = n = 0
            for m in range(13 if flat else 14):
                for tap in range(9):
                    tapoff = (tap // 3) * 16 + tap % 3
                    for ks in range(4):
                        tot += cycles_b128(conv_x_read(stride, m, tapoff, ks, flat)); n += 1
            print('stride', stride, 'flat' if flat else 'rows', 'avg cycles per read %.2f (ideal 4)' % (tot / n))

def conv_x_read_pitch(stride, m, tapoff, ks, pitch):
    addrs = []
    for l in range(64):
        l15, lq = l & 15, l >> 4
        px = min(m * 16 + l15, 195); y, x = divmod(px, 14)
        addrs.append((y * pitch + x + tapoff) * stride + lq * 16 + ks * 64)
    return addrs

def sweep_pitch():
    for pitch in (15, 16):
        for stride in range(256, 417, 16):
            tot = n = 0
            for m in range(13):
                for tap in range(9):
                    tapoff = (tap // 3) * pitch + tap % 3
                    for ks in range(4):
                        tot += cycles_b128(conv_x_read_pitch(stride, m, tapoff, ks, pitch)); n += 1
            print('pitch', pitch, 'stride', stride, 'avg %.2f' % (tot / n))
