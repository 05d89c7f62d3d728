import sys, json
for l in sys.stdin:
    j = json.loads(l)
    print(j["fft"], j["hop"], round(j["ms"], 3), round(j["frac_of_8TBps"], 3))
