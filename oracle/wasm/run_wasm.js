// TEST INFRASTRUCTURE ONLY (oracle): drives the reference's own shipped WebAssembly build
// (web/emscripten/main.js:9 of the reference tree = the real signalsmith-stretch.h + the real
// signalsmith-linear, -O3 -ffast-math, float) under Node >= 12.  Used in the build container only
// (the reference tree does not travel to the GPU box) to generate tests/golden/*.
//
// usage: node run_wasm.js <main.js> <job.json>
// job: { channels, sampleRate, preset: "default"|"cheaper"|"configure", block, interval, split,
//        input: "<planar f32 file, channels x inTotal>", inTotal, output: "<planar f32 file>",
//        ops: [ {op:"setTransposeSemitones", args:[..]} | {op:"setTransposeFactor",args} |
//               {op:"setFormantSemitones",args} | {op:"setFormantFactor",args} | {op:"setFormantBase",args} |
//               {op:"reset"} | {op:"seek", inStart, inLen, rate} |
//               {op:"process", inStart, inLen, outLen} | {op:"flush", outLen} ] }
// Output file: planar, channels x (sum of outLen over process/flush ops), ops concatenated in time.
// Export letters (declaration order of web/emscripten/main.cpp:16-76):
//  h setBuffers i blockSamples j intervalSamples k inputLatency l outputLatency m reset n presetDefault
//  o presetCheaper p configure q setTransposeFactor r setTransposeSemitones s setFormantFactor
//  t setFormantSemitones u setFormantBase v seek w process x flush
'use strict';
const fs = require('fs');
const src = fs.readFileSync(process.argv[2], 'utf8');
const job = JSON.parse(fs.readFileSync(process.argv[3], 'utf8'));
const m = /data:application\/octet-stream;base64,([A-Za-z0-9+\/=]+)/.exec(src);
if (!m) { console.error('no embedded wasm found'); process.exit(2); }
const bin = Buffer.from(m[1], 'base64');
let mem;
const inst = new WebAssembly.Instance(new WebAssembly.Module(bin), {a: {
  a: (p, n) => { new Uint8Array(mem.buffer).fill(7, p, p + n); return 0; },
  b: (req) => { try { mem.grow(Math.ceil(((req >>> 0) - mem.buffer.byteLength) / 65536)); return 1; } catch (e) { return 0; } },
  c: (d, s, n) => new Uint8Array(mem.buffer).copyWithin(d, s, s + n),
  d: () => { throw new Error('abort'); }}});
const X = inst.exports; mem = X.e; X.f();
const C = job.channels;
if (job.preset === 'default') X.n(C, job.sampleRate);
else if (job.preset === 'cheaper') X.o(C, job.sampleRate);
else X.p(C, job.block, job.interval, job.split ? 1 : 0);
const info = {block: X.i(), interval: X.j(), inputLatency: X.k(), outputLatency: X.l()};
const inTotal = job.inTotal;
const inBuf = inTotal > 0 ? fs.readFileSync(job.input) : Buffer.alloc(0);
const inAll = new Float32Array(inBuf.buffer, inBuf.byteOffset, C * inTotal);
let maxLen = 1;
for (const op of job.ops) maxLen = Math.max(maxLen, op.inLen || 0, op.outLen || 0);
const ptr = X.h(C, maxLen);
let totalOut = 0;
for (const op of job.ops) if (op.op === 'process' || op.op === 'flush') totalOut += op.outLen;
const outAll = new Float32Array(C * totalOut);
let outPos = 0;
const names = {setTransposeFactor: 'q', setTransposeSemitones: 'r', setFormantFactor: 's', setFormantSemitones: 't', setFormantBase: 'u'};
const t0 = Date.now();
let procMs = 0;
for (const op of job.ops) {
  if (names[op.op]) { X[names[op.op]].apply(null, op.args); continue; }
  if (op.op === 'reset') { X.m(); continue; }
  let f32 = new Float32Array(mem.buffer);
  if (op.op === 'seek' || op.op === 'process') {
    for (let c = 0; c < C; ++c) {
      f32.set(inAll.subarray(c * inTotal + op.inStart, c * inTotal + op.inStart + op.inLen), (ptr >> 2) + maxLen * c);
    }
  }
  const t1 = Date.now();
  if (op.op === 'seek') { X.v(op.inLen, op.rate); continue; }
  if (op.op === 'process') X.w(op.inLen, op.outLen);
  else if (op.op === 'flush') X.x(op.outLen);
  else { console.error('bad op ' + op.op); process.exit(2); }
  procMs += Date.now() - t1;
  f32 = new Float32Array(mem.buffer);
  for (let c = 0; c < C; ++c) {
    outAll.set(f32.subarray((ptr >> 2) + maxLen * (C + c), (ptr >> 2) + maxLen * (C + c) + op.outLen), c * totalOut + outPos);
  }
  outPos += op.outLen;
}
fs.writeFileSync(job.output, Buffer.from(outAll.buffer));
if (job.dumpMemory) fs.writeFileSync(job.dumpMemory, Buffer.from(mem.buffer)); // linear memory after the last op (state probes, SURVEY.md App. B)
info.totalOut = totalOut; info.processMs = procMs;
console.log(JSON.stringify(info));
