"""Expected `CHECK` lines of tools/shake_dev.hip from hashlib.shake_128 (same messages: 16 key bytes + b'uci')."""
import hashlib, struct
B = 1000
for s in (0, 19):
    msg = bytes((7 * s + j) & 0xff for j in range(16)) + b'uci'
    d = hashlib.shake_128(msg).digest(B * 168)
    x = 0
    for i, (w,) in enumerate(struct.iter_unpack('<Q', d)):
        x ^= (w * (2 * i + 1)) & (2**64 - 1)
    print(f'CHECK stream {s} first16 {d[:16].hex()} fold {x:016x}')
