# configs[2] through the driver's multi-device paths on ONE GPU (--devices 0,0,0: three worker contexts on device 0): compress and decompress, files on /dev/shm.
# The rates mean little (three contexts share a device and one PCIe link); what is shown is that both chunk-parallel paths carry the full-size input:
# md5 of the image == the reference's, both mates byte-identical after -d --devices.   usage: bash tools/e2e_pe_devices.sh
set -e
cd $GRAFT_REPO_ROOT
D=/dev/shm/e2e; mkdir -p $D
./tools/fqgen --profile 1 --reads 11200000 --seed 3 -o $D/r1.fq -O $D/r2.fq
B=repaq_amd/bin/repaq_hip
TIMEFORMAT="PE compress --devices 0,0,0 wall %R s"; time $B -c -i $D/r1.fq -I $D/r2.fq -o $D/pe.rfq --devices 0,0,0
md5sum $D/pe.rfq
TIMEFORMAT="PE decompress --devices 0,0,0 wall %R s"; time $B -d -i $D/pe.rfq -o $D/o1.fq -O $D/o2.fq --devices 0,0,0
cmp $D/r1.fq $D/o1.fq && cmp $D/r2.fq $D/o2.fq && echo PE_DEVICES_ROUNDTRIP_OK
TIMEFORMAT="PE decompress (one device) wall %R s"; time $B -d -i $D/pe.rfq -o $D/o1.fq -O $D/o2.fq
cmp $D/r1.fq $D/o1.fq && cmp $D/r2.fq $D/o2.fq && echo PE_ROUNDTRIP_OK
rm -rf /dev/shm/e2e
