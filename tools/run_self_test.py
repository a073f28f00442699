import sys; sys.path.insert(0,'/root/repo')
import raytracingrenderer_amd as rt
print("self test mismatches:", rt.self_test(0))
