import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
h = b'\xe0\xa4\xa8\xd9\x86\xc3\xa9_\xd9\x86\xd9\x86\xf0\x9f\x98\x80\xc3\xa9\xe0\xa4\xa8\xc3\xa9\xc3\xa9\xf0\x9f\x98\x80\xd9\x86_ \xeb\x8b\xa4\xf0\x9f\x98\x80\xe0\xa4\xa8\xeb\x8b\xa4_\xf0\x9f\x98\x80\xeb\x8b\xa4\xe0\xa4\xa8\xc3\xa9_\xd9\x86 \xf0\x9f\x98\x80 \xf0\x9f\x98\x80\xf0\x9f\x98\x80'
if len(sys.argv) > 1:
    import frizbee_amd as F
    k, pf, cut = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    m = F.Matcher('😀😀ن', F.Config(max_typos=k, pf_lanes=pf, sw_lanes=pf))
    hh = h[:cut].decode('utf-8', 'ignore').encode()
    print(k, pf, len(hh), m.match_list([hh]).tolist())
    sys.exit(0)
for lib in ("", "O1"):
    env = dict(os.environ)
    if lib: env["FRIZBEE_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "frizbee_amd", "libfrizbee_hip_O1.so")
    for k, pf, cut in [(2, 64, 76), (1, 64, 76), (3, 64, 76), (2, 32, 76), (2, 16, 76), (2, 64, 64), (2, 64, 60), (2, 64, 40), (2, 64, 70)]:
        try:
            p = subprocess.run([sys.executable, __file__, str(k), str(pf), str(cut)], capture_output=True, text=True, timeout=15, env=env)
            print(lib or "O3", p.stdout.strip(), p.stderr.strip()[-200:], flush=True)
        except subprocess.TimeoutExpired:
            print(lib or "O3", k, pf, cut, "TIMEOUT", flush=True)
