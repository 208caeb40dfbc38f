import json,sys
for f in sys.argv[1:]:
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f, d['ms_per_step'], d['roofline']['frac'])
    for view,rows in d['gui_latency'].items():
        if isinstance(rows,dict):
            for k,v in rows.items():
                if isinstance(v,dict) and 'rgb' in v: print('  ',view[:20],k[:18],'rgb first %.3f median %.3f | rgba first %.3f median %.3f'%(v['rgb']['first_call_ms'],v['rgb']['median_ms'],v['rgba']['first_call_ms'],v['rgba']['median_ms']))
