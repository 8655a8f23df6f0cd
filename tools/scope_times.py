import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
o=d['roofline'].get('others',{})
print(sys.argv[1], d['ms_per_step'])
for k in ('conv3d_m2_64_32','dgrad_m2_64_32','conv3d_m1_32_64','dgrad_m1_32_64','conv_wgrad_s2_64_32','deconv_wgrad_s2_64_32','conv_wgrad_s1_32_32','dgrad_m0_32_32','conv3d_m0_32_32','bn3d_bwd_32','bn3d_bwd_64'):
    if k in o: print("  %-26s %.3f ms x %d"%(k,o[k]['avg_ms'],o[k]['launches']))
    elif d['roofline'].get('kernel')==k: print("  %-26s %.3f ms (top)"%(k,d['roofline']['avg_launch_ms']))
