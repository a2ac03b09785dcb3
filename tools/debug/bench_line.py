import json,sys
for l in open(sys.argv[1]):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],4), 'vox', round(d['roofline_voxel']['stage_ms'],5), 'conv', d['roofline'].get('launch_ms'))
