import re,sys
txt=open(sys.argv[1]).read()
for line in txt.split('\n'):
    if not line.startswith('part'): continue
    part=int(line[5])
    ev=re.findall(r'([a-zA-Z0-9 ]+?) (\d+\.\d+)(?:  |$)', line[8:])
    d={k.strip():float(v) for k,v in ev}
    f=0.0417
    t0=d['start']
    print("part",part, "phase0 %.1f"%((d['gathered']-t0)*f), "own %.1f (conv %.1f, steps %.1f %.1f %.1f %.1f, Tw %.1f)"%((d['T written']-d['own begin'])*f,(d['converted']-d['own begin'])*f,(d['step 8']-d['converted'])*f,(d['step 16']-d['step 8'])*f,(d['step 24']-d['step 16'])*f,(d['step 32']-d['step 24'])*f,(d['T written']-d['step 32'])*f), "total %.1f"%((d['end']-t0)*f))
