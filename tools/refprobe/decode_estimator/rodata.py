import struct, sys
data = open('/root/reference/cassie/cassiemujoco/libcassiemujoco.so','rb').read()
def rd(a): return struct.unpack('<d', data[a:a+8])[0]
if __name__ == "__main__":
    for a in sys.argv[1:]:
        a = int(a, 16); print(hex(a), rd(a), data[a:a+8].hex())
