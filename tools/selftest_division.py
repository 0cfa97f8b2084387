import sys, ctypes; sys.path.insert(0,'/root/repo')
from sph_amd import lib
L=lib.load()
c=(ctypes.c_ulonglong*8)(); m=(ctypes.c_ulonglong*8)()
lib.check(L.sphhip_selftest_division(0,c,m))
print("checked", list(c)[:7]); print("mismatches", list(m)[:7])
