import json,sys
d=json.load(open(sys.argv[1]))
f=d['fitting']
print({k:(round(v.get('ms_per_step'),3) if isinstance(v,dict) and v.get('ms_per_step') else None) for k,v in f.items()})
print('fps', {k:(round(v.get('frames_per_s'),3)) for k,v in f.items() if isinstance(v,dict) and v.get('frames_per_s')})
print(d['value']/1e6, d['roofline'].get('frac'), d['ms_per_step'])
