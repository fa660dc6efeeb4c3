import sys, json
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l)
        print(d['value'], d['ms_per_step'], d.get('bracketed_ms_per_step'))
        for k, v in d.get('kernel_classes', {}).items():
            print(k, v)
