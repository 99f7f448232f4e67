"""agpr_attr.py IN.ll OUT.ll "REGEX=N[,REGEX=N...]" -- give the device functions whose name matches REGEX the LLVM function attribute
"amdgpu-agpr-alloc"="N" (see hipcc_agpr.sh).  Each matching definition gets a fresh attribute group: a copy of its own plus the attribute."""
import re
import sys


def main():
    src, dst, spec = sys.argv[1:4]
    rules = [(re.compile(r.split('=')[0]), int(r.split('=')[1])) for r in spec.split(',') if r]
    text = open(src).read()
    groups = dict(re.findall(r'^attributes #(\d+) = \{(.*)\}\s*$', text, flags=re.M))
    next_id = max(int(g) for g in groups) + 1
    added, hits = [], []

    def patch(m):
        nonlocal next_id
        name, gid = m.group(2), m.group(3)
        for rx, n in rules:
            if rx.search(name):
                body = re.sub(r'\s*"amdgpu-agpr-alloc"="[^"]*"', '', groups[gid])
                added.append('attributes #%d = {%s "amdgpu-agpr-alloc"="%d" }' % (next_id, body.rstrip(), n))
                hits.append((name, n))
                out = '%s#%d%s' % (m.group(1), next_id, m.group(4))
                next_id += 1
                return out
        return m.group(0)

    # define ... @name(args) [unnamed_addr] #G [!metadata ...] {
    text = re.sub(r'^(define [^\n]*@([\w.$]+)\([^\n]*?\)[^\n#]*)#(\d+)([^\n]*\{)\s*$', patch, text, flags=re.M)
    if not hits:
        sys.exit('agpr_attr.py: no function matches %r' % spec)
    open(dst, 'w').write(text + '\n' + '\n'.join(added) + '\n')
    for name, n in hits:
        print('agpr_attr: %s -> amdgpu-agpr-alloc=%d' % (name, n))


if __name__ == '__main__':
    main()
